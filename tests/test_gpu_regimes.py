"""-m gpu: the loss on emissions OTHER than unit-variance noise, at the headline's lattice size (T=1000, S<=200), against the
oracle: what a trained model emits (peaky, consistent with the targets), mislabelled utterances in front of such a model,
and sharp emissions that have nothing to do with the targets.  The last two defeat any lattice scaled by one power of
two per row; they are settled inside the same call by the extended-range redo (end2end_amd/csrc/ctc_ext.h), not by the
exact kernel's full recomputation -- asserted through the call's diagnostics."""
import ctypes

import numpy as np
import pytest
import torch

import oracle_lib as O
import gpu_util as U
from end2end_amd import _lib

pytestmark = pytest.mark.gpu

F32_RTOL, F32_ATOL = 1e-4, 2e-6


def aligned(rng, B, T, V, S, boost, blank=0, short=()):
    """unit noise + `boost` on one random monotone alignment of each utterance's own targets (bench.py aligned_batch);
    short: (utterance, frames) pairs -- utterances that end early (their alignment lies inside their own frames)"""
    x = rng.standard_normal((B, T, V)).astype(np.float32)
    tg = rng.integers(1, V, size=(B, S))
    tl = rng.integers(max(S // 2, 1), S + 1, size=B)
    xl = np.full(B, T)
    for b, frames in short:
        xl[b] = frames
    for b in range(B):
        n, Tb = int(tl[b]), int(xl[b])
        slots = np.sort(rng.choice(Tb, size=n, replace=False))
        path = np.full(Tb, blank)
        path[slots] = tg[b, :n]
        clash = np.nonzero((tg[b, 1:n] == tg[b, :n - 1]) & (slots[1:] == slots[:-1] + 1))[0] + 1
        path[slots[clash]] = blank
        x[b, np.arange(Tb), path] += boost
    return x, tg, xl, tl


def unsettled(keep, B, T, V, S):
    """utterances the exact kernel had to recompute in full; also: no bounded wait of the flagged launch may have run out"""
    L = _lib.load()
    L.e2e_debug_fast_redo_failures.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    L.e2e_debug_flagged_counters.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
    cnt, to, fr = ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_int(-1)
    assert L.e2e_debug_fast_redo_failures(keep["workspace"].data_ptr(), B, T, V, S, ctypes.byref(cnt)) == 0
    assert L.e2e_debug_flagged_counters(keep["workspace"].data_ptr(), B, T, V, S, ctypes.byref(to), ctypes.byref(fr)) == 0
    assert to.value == 0, "%d bounded waits of the flagged launch ran out" % to.value
    keep["failed_redos"] = fr.value
    return cnt.value


def check(x, tg, xl, tl, want_unsettled=0, loss_atol=2e-5, keep=None):
    B, T, V = x.shape
    S = tg.shape[1]
    xt = torch.from_numpy(x)
    lp = torch.log_softmax(xt.double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg, xl, tl, 0)
    for b in range(B):
        g_o[b, xl[b]:] = 0.0
    keep = {} if keep is None else keep
    la, ga = U.c_abi_loss(xt, tg, xl, tl, 0, False, _lib.ALGO_AUTO, keep=keep)
    U.assert_same(la, l_o, F32_RTOL, loss_atol, "losses")
    U.assert_same(ga, g_o, F32_RTOL, F32_ATOL, "grads")
    if want_unsettled is not None:
        assert unsettled(keep, B, T, V, S) == want_unsettled
    return l_o


@pytest.mark.parametrize("boost", [6.0, 10.0, 14.0])
def test_trained_regime_at_the_headline_size_stays_on_the_fast_path(boost):
    rng = np.random.default_rng(int(boost))
    x, tg, xl, tl = aligned(rng, 6, 1000, 29, 200, boost, short=[(2, 871)])
    lf, _ = U.c_abi_loss(torch.from_numpy(x), tg, xl, tl, 0, False, _lib.ALGO_FAST)
    assert not np.isnan(lf).any(), "the fast path gave up on %s" % np.nonzero(np.isnan(lf))[0].tolist()
    check(x, tg, xl, tl)


def test_mislabelled_utterances_are_settled_in_extended_range():
    rng = np.random.default_rng(21)
    x, tg, xl, tl = aligned(rng, 8, 1000, 29, 200, 10.0, short=[(5, 933)])
    tg[1], tl[1] = tg[2].copy(), tl[2]              # utterance 1 carries utterance 2's transcript, utterance 5 utterance 6's
    tg[5], tl[5] = tg[6].copy(), tl[6]
    lf, _ = U.c_abi_loss(torch.from_numpy(x), tg, xl, tl, 0, False, _lib.ALGO_FAST)
    assert np.isnan(lf[[1, 5]]).all() and np.isnan(lf).sum() == 2, "expected exactly the mislabelled utterances to leave the f32 lattice"
    l_o = check(x, tg, xl, tl)
    assert l_o[1] > 20 * l_o[0]


@pytest.mark.parametrize("shape", [(6, 1000, 29, 200, 8.0), (4, 500, 29, 100, 8.0), (3, 2000, 29, 400, 3.0), (3, 1000, 80, 200, 8.0),
                                   (3, 700, 29, 447, 8.0), (3, 600, 150, 150, 8.0), (2, 512, 300, 200, 8.0)],
                         ids=lambda s: "B%d_T%d_V%d_S%d_x%g" % s)
def test_sharp_unrelated_emissions_are_settled_in_extended_range(shape):
    B, T, V, S, scale = shape
    rng = np.random.default_rng(B + T)
    x = (rng.standard_normal((B, T, V)) * scale).astype(np.float32)
    tg = rng.integers(1, V, size=(B, S))
    tl = rng.integers(S // 2, S + 1, size=B)
    tl[0] = S
    xl = np.full(B, T)
    xl[1] = T - 21
    check(x, tg, xl, tl)


def test_logprob_input_with_impossible_symbols_in_extended_range():
    """log-probabilities with -inf entries (symbols that cannot be emitted at a frame) under sharp unrelated emissions: the
    extended-range cells treat an exact zero as such."""
    rng = np.random.default_rng(4)
    B, T, V, S = 4, 400, 20, 60
    x = rng.standard_normal((B, T, V)) * 8.0
    x[:, ::7, 3] = -np.inf
    lp = torch.log_softmax(torch.from_numpy(x), -1)
    tg = rng.integers(1, V, size=(B, S)); tl = rng.integers(S // 2, S + 1, size=B); xl = np.full(B, T)
    l_o, g_o = O.ctc_loss(lp.numpy(), tg, xl, tl, 0)
    la, ga = U.c_abi_loss(lp.float(), tg, xl, tl, 0, True, _lib.ALGO_AUTO)
    U.assert_same(la, l_o, F32_RTOL, 2e-5, "losses")
    U.assert_same(ga, g_o, F32_RTOL, F32_ATOL, "grads")


@pytest.mark.parametrize("V", [40, 150], ids=["segment_table_V40", "row_table_V150"])
def test_probabilities_at_the_end_of_f32_go_where_they_can_be_held(V):
    """Sharp unrelated emissions with, on top, a frame whose blank -- which nearly every path of a three-label transcript crosses --
    has log-probability -74 (utterance 1: below 2^-100, flag 64, but a normal f32 number: the extended-range redo takes it from the
    fast path's table) or -82 (utterance 0: the table holds a marker instead, flag 256: only the exact kernel's own softmax can
    serve it).  Round 5 shipped this for an hour with the 256 vote taken inside `if (lane == 0)`: the marker was then used as a
    probability and the loss came out 4.4 nats off with perfect gradients (tools/diag/fuzz_ext_vs_oracle.py, case 196 of seed 3)."""
    rng = np.random.default_rng(V)
    B, T, S = 4, 400, 3
    x = (rng.standard_normal((B, T, V)) * 8.0).astype(np.float32)
    x[0, 100, 0] = x[0, 100].max() - 82.0
    x[1, 100, 0] = x[1, 100].max() - 74.0
    x[1, 300, 0] = x[1, 300].max() - 71.0
    tg = rng.integers(1, V, size=(B, S)); tl = np.full(B, S); xl = np.array([T, T, T - 13, T])
    check(x, tg, xl, tl, want_unsettled=None, loss_atol=2e-5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_sixteen_bit_logits_through_the_extended_range_redo(dtype):
    """16-bit logits (read natively, gradient written in the same type, losses f32) with sharp unrelated emissions: the flagged
    launch's 16-bit instances carry the extended-range redo too.  Against the oracle on the rounded logits, at the output type's
    resolution."""
    rng = np.random.default_rng(12)
    B, T, V, S = 4, 500, 29, 100
    x16 = torch.from_numpy((rng.standard_normal((B, T, V)) * 8.0).astype(np.float32)).to(dtype)
    tg = rng.integers(1, V, size=(B, S)); tl = rng.integers(S // 2, S + 1, size=B); xl = np.array([T, T - 30, T, 411])
    lp = torch.log_softmax(x16.double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg, xl, tl, 0)
    for b in range(B):
        g_o[b, xl[b]:] = 0.0
    keep = {}
    lf, _ = U.c_abi_loss(x16, tg, xl, tl, 0, False, _lib.ALGO_FAST)
    assert np.isnan(lf).sum() >= 2, "this input no longer drives utterances off the f32 lattice"
    la, ga = U.c_abi_loss(x16, tg, xl, tl, 0, False, _lib.ALGO_AUTO, keep=keep)
    eps = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    U.assert_same(la, l_o, 1e-4, 2e-5, "losses")
    U.assert_same(ga, g_o, 2 * eps, 2e-6, "grads")
    assert unsettled(keep, B, T, V, S) == 0


def test_a_failed_segment_redo_is_settled_by_the_second_round():
    """Logits of scale 4 against unrelated transcripts: flagged for range only, so the f64 redo of single segments runs first --
    and cannot hold some rows (a segment's lattice spans more than f64's exponent range).  The flagged launch learns that at its
    bounded wait and settles those utterances in extended range (round 1: every workgroup releases what the redo wrote before
    the round rewrites the rows).  Nothing may be left to the full recomputation."""
    B, T, V, S = 8, 1000, 29, 200              # (tools/diag/find_round1.py: which inputs take this path)
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((B, T, V)) * 4.0).astype(np.float32)
    tg = rng.integers(1, V, size=(B, S)); tl = rng.integers(S // 2, S + 1, size=B); xl = np.full(B, T)
    keep = {}
    check(x, tg, xl, tl, keep=keep)
    assert keep["failed_redos"] > 0, "this input no longer makes a segment redo fail: the second round is not exercised"


@pytest.mark.parametrize("scale,seed", [(3.0, 5), (4.0, 0)], ids=["no_redo_fails", "some_redo_fails"])
def test_range_flags_beside_an_utterance_flagged_for_its_inputs(scale, seed):
    """The flagged launch's bounded wait with work left for step 2: utterances flagged for range only (their segments are redone in
    f64; at scale 4 some redos fail and the second round settles them) in one batch with a blank-valued target, which only the
    reference's arithmetic serves.  With such an utterance present the workgroup that arrives last at the wait must NOT take
    the reduction early (another one still has step 2 to do): the fused sum of the losses is checked for that."""
    B, T, V, S = 9, 1000, 29, 200
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((B, T, V)) * scale).astype(np.float32)
    tg = rng.integers(1, V, size=(B, S)); tl = rng.integers(S // 2, S + 1, size=B); xl = np.full(B, T)
    tg[8, 5] = 0                                   # a target equal to the blank id
    xt = torch.from_numpy(x)
    lp = torch.log_softmax(xt.double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg, xl, tl, 0)
    lf, _ = U.c_abi_loss(xt, tg, xl, tl, 0, False, _lib.ALGO_FAST)
    assert np.isnan(lf[:8]).sum() >= 2 and np.isnan(lf[8]), "this input no longer drives utterances off the f32 lattice"
    keep = {}
    la, ga, red = U.c_abi_loss(xt, tg, xl, tl, 0, False, _lib.ALGO_AUTO, keep=keep, opts=(1.0, _lib.REDUCE_SUM))
    U.assert_same(la, l_o, F32_RTOL, 2e-5, "losses")
    U.assert_same(ga, g_o, F32_RTOL, F32_ATOL, "grads")
    assert abs(red - l_o.sum()) <= 1e-5 * abs(l_o.sum())
    assert unsettled(keep, B, T, V, S) == 0
    assert (keep["failed_redos"] > 0) == (scale == 4.0)


def test_more_flagged_utterances_than_spare_workgroups_keep_both_directions_together():
    """With at most 128 utterances for the extended-range redo, the alpha and the beta chains of an utterance run on two workgroups
    of the flagged launch (round 6); beyond, on one, as before.  160 short utterances of sharp unrelated emissions take the second
    route (every other test of this file, and the fuzz slices, take the first)."""
    rng = np.random.default_rng(160)
    B, T, V, S = 160, 96, 29, 24
    x = (rng.standard_normal((B, T, V)) * 8.0).astype(np.float32)
    tg = rng.integers(1, V, size=(B, S))
    tl = rng.integers(S // 2, S + 1, size=B)
    xl = rng.integers(T // 2 + S, T + 1, size=B)
    xl[0] = T
    lf, _ = U.c_abi_loss(torch.from_numpy(x), tg, xl, tl, 0, False, _lib.ALGO_FAST)
    assert np.isnan(lf).sum() > 128, "only %d utterances left the f32 lattice: the test needs more than 128" % np.isnan(lf).sum()
    check(x, tg, xl, tl, want_unsettled=None)
