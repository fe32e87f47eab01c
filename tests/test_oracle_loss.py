"""CPU: the oracle's loss/grad against the reference's golden vectors and engine outputs."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O


def same(a, b, rtol, atol):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b)
    inf_a = np.isinf(a)
    assert np.array_equal(inf_a, np.isinf(b))
    assert np.array_equal(a[inf_a], b[inf_a])
    ok = ~(nan_a | inf_a)
    np.testing.assert_allclose(a[ok], b[ok], rtol=rtol, atol=atol)


def test_lse_matches_math_utils():
    L = O.lib()
    assert L.oracle_log_sum_exp(-np.inf, -3.0) == -3.0
    assert L.oracle_log_sum_exp(-3.0, -np.inf) == -3.0
    assert L.oracle_log_sum_exp(-np.inf, -np.inf) == -np.inf
    assert abs(L.oracle_log_sum_exp(np.log(0.25), np.log(0.5)) - np.log(0.75)) < 1e-15


@pytest.mark.parametrize("case", G.known_answers()["loss"], ids=lambda c: c["name"])
def test_known_answer_costs(case):
    # reference tests assert 5 decimal places on the summed cost (tests/test_ctc.py:19,75-77)
    lp, tg, xl, tl, blank, cost = G.known_loss_inputs(case)
    losses, _ = O.ctc_loss(lp, tg, xl, tl, blank)
    assert round(abs(losses.sum() - cost), 5) == 0


@pytest.mark.parametrize("m", G.meta()["engine"], ids=lambda m: m["name"])
def test_engine_fixtures(m):
    c = G.engine_case(m["name"])
    losses, grads = O.ctc_loss(c["lp"], c["targets"], c["x_len"], c["t_len"], m["blank"])
    if m["dtype"] == "float64":
        same(losses, c["losses"], 1e-13, 1e-13)
        same(grads, c["grads"], 1e-12, 1e-14)
    else:  # the reference rounds its f64 result to f32 once (forward_backward.cpp:55-56)
        same(losses.astype(np.float32), c["losses"], 0, 0)
        same(grads.astype(np.float32), c["grads"], 1e-6, 1e-9)


def test_padded_rows_are_softmax_and_infeasible_is_nan():
    c = G.engine_case("ragged_f32")
    _, grads = O.ctc_loss(c["lp"], c["targets"], c["x_len"], c["t_len"], 0)
    for b, n in enumerate(c["x_len"]):
        np.testing.assert_allclose(grads[b, n:], np.exp(c["lp"][b, n:].astype(np.float64)), rtol=1e-12)
    e = G.engine_case("edges_f32")
    losses, grads = O.ctc_loss(e["lp"], e["targets"], e["x_len"], e["t_len"], 0)
    assert np.isinf(losses[3]) and losses[3] > 0 and np.isnan(grads[3]).all()


def test_oracle_vs_live_reference_engine():
    ref = O.load_reference_engine()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    import torch
    g = torch.Generator().manual_seed(11)
    lp = torch.log_softmax(torch.randn(5, 37, 11, generator=g, dtype=torch.float64), -1)
    tg = torch.randint(1, 11, (5, 9), generator=g)
    xl = torch.tensor([37, 30, 25, 37, 19])
    tl = torch.tensor([9, 4, 0, 7, 9])
    l_ref, g_ref = ref.CTCLossEngine(0).compute(lp, tg, xl, tl)
    l_o, g_o = O.ctc_loss(lp.numpy(), tg.numpy(), xl.numpy(), tl.numpy(), 0)
    same(l_o, l_ref.numpy(), 1e-14, 0)
    same(g_o, g_ref.numpy(), 1e-12, 1e-15)


def test_thread_per_utterance_equals_pool():
    c = G.engine_case("long_f32")
    a = O.ctc_loss(c["lp"], c["targets"], c["x_len"], c["t_len"], 0, n_threads=0)
    b = O.ctc_loss(c["lp"], c["targets"], c["x_len"], c["t_len"], 0, n_threads=2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1], equal_nan=True)
