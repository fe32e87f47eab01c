"""-m gpu: HIP CTC loss/grad through the C ABI against the golden fixtures and the oracle.

Tolerances: north_star asks for 1e-4 relative in fp32; f64 inputs must track the reference's f64
arithmetic (gradcheck, tests/test_ctc.py:168-191) and are held to 1e-9.
"""
import numpy as np
import pytest
import torch

import golden_util as G
import oracle_lib as O
import gpu_util as U
from end2end_amd import _lib

pytestmark = pytest.mark.gpu

F32_RTOL, F32_ATOL = 1e-4, 2e-6
F64_RTOL, F64_ATOL = 1e-9, 1e-12
ALGOS = [_lib.ALGO_AUTO, _lib.ALGO_EXACT, _lib.ALGO_FAST]
ALGO_IDS = {_lib.ALGO_AUTO: "auto", _lib.ALGO_EXACT: "exact", _lib.ALGO_FAST: "fast"}


def run(x, tg, xl, tl, blank, logprobs, algo, want_losses):
    """Run one algorithm.  ALGO_FAST has no fallback: utterances it cannot do (infeasible alignment, out-of-range
    lattice) come back NaN-poisoned; those must be exactly the ones the oracle calls infeasible, and are then taken
    from the auto path so that the caller compares full tensors."""
    if algo == _lib.ALGO_FAST and x.dtype != torch.float32:
        pytest.skip("the fast path is f32 only")
    losses, grads = U.c_abi_loss(x, tg, xl, tl, blank, logprobs, algo)
    if algo == _lib.ALGO_FAST:
        flagged = np.isnan(losses)
        assert np.array_equal(flagged, ~np.isfinite(np.asarray(want_losses, dtype=np.float64))), \
            "fast path flagged %s" % np.nonzero(flagged)[0].tolist()
        if flagged.any():
            assert np.isnan(grads[flagged]).all()
            l2, g2 = U.c_abi_loss(x, tg, xl, tl, blank, logprobs, _lib.ALGO_AUTO)
            losses[flagged], grads[flagged] = l2[flagged], g2[flagged]
    return losses, grads


@pytest.mark.parametrize("algo", ALGOS, ids=ALGO_IDS.get)
@pytest.mark.parametrize("m", G.meta()["engine"], ids=lambda m: m["name"])
def test_engine_fixtures(m, algo):
    c = G.engine_case(m["name"])
    lp = torch.from_numpy(c["lp"])
    if m["name"].startswith("permuted_view"):
        lp = lp.permute(1, 0, 2).contiguous().permute(1, 0, 2)     # non-contiguous (B,T,V) view
    losses, grads = run(lp, c["targets"], c["x_len"], c["t_len"], m["blank"], True, algo, c["losses"])
    rt, at = (F64_RTOL, F64_ATOL) if m["dtype"] == "float64" else (F32_RTOL, F32_ATOL)
    U.assert_same(losses, c["losses"], rt, at, m["name"] + " losses")
    U.assert_same(grads, c["grads"], rt, at, m["name"] + " grads")


@pytest.mark.parametrize("algo", ALGOS, ids=ALGO_IDS.get)
@pytest.mark.parametrize("case", G.known_answers()["loss"], ids=lambda c: c["name"])
def test_known_answer_costs(case, algo):
    lp, tg, xl, tl, blank, cost = G.known_loss_inputs(case)
    losses, _ = run(torch.from_numpy(lp.astype(np.float32)), tg, xl, tl, blank, True, algo, [0.0] * len(xl))
    assert round(abs(float(losses.astype(np.float64).sum()) - cost), 5) == 0


@pytest.mark.parametrize("algo", ALGOS, ids=ALGO_IDS.get)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_fused_logits_equals_logsoftmax_then_engine(dtype, algo):
    g = torch.Generator().manual_seed(21)
    B, T, V, S = 5, 61, 13, 14
    x = torch.randn(B, T, V, generator=g, dtype=torch.float64).to(dtype)
    tg = torch.randint(1, V, (B, S), generator=g)
    xl = torch.tensor([61, 50, 33, 61, 29])
    tl = torch.tensor([14, 9, 0, 11, 14])
    lp = torch.log_softmax(x.double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg.numpy(), xl.numpy(), tl.numpy(), 0)
    losses, grads = run(x, tg, xl, tl, 0, False, algo, l_o)
    for b in range(B):
        g_o[b, xl[b]:] = 0.0            # what autograd through log_softmax leaves on padded frames
    rt, at = (F64_RTOL, F64_ATOL) if dtype == torch.float64 else (F32_RTOL, F32_ATOL)
    U.assert_same(losses, l_o, rt, at, "losses")
    U.assert_same(grads, g_o, rt, at, "grads")


@pytest.mark.parametrize("algo", ALGOS, ids=ALGO_IDS.get)
def test_c2_shape_sample_against_oracle(algo):
    # the headline shape (T=1000, V=29, S in [100,200]) on a few utterances the oracle finishes in seconds
    g = torch.Generator().manual_seed(0)
    B, T, V, S = 6, 1000, 29, 200
    x = torch.randn(B, T, V, generator=g)
    tg = torch.randint(1, V, (B, S), generator=g)
    tl = torch.randint(S // 2, S + 1, (B,), generator=g)
    tl[0] = S
    xl = torch.tensor([T, T, T - 37, T, 640, T])
    lp = torch.log_softmax(x.double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg.numpy(), xl.numpy(), tl.numpy(), 0)
    losses, grads = run(x, tg, xl, tl, 0, False, algo, l_o)
    for b in range(B):
        g_o[b, xl[b]:] = 0.0
    U.assert_same(losses, l_o, F32_RTOL, F32_ATOL, "losses")
    U.assert_same(grads, g_o, F32_RTOL, F32_ATOL, "grads")


def test_c2_shape_sample_with_f32_chains_against_oracle():
    # the same with e2e_ctc_loss_opts.chains = E2E_CHAINS_F32: losses as tight as ever, gradient elements within the
    # 2e-5 absolute that include/e2e_ctc.h states for the option
    g = torch.Generator().manual_seed(0)
    B, T, V, S = 6, 1000, 29, 200
    x = torch.randn(B, T, V, generator=g)
    tg = torch.randint(1, V, (B, S), generator=g)
    tl = torch.randint(S // 2, S + 1, (B,), generator=g)
    tl[0] = S
    xl = torch.tensor([T, T, T - 37, T, 640, T])
    lp = torch.log_softmax(x.double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg.numpy(), xl.numpy(), tl.numpy(), 0)
    losses, grads = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_FAST, chains=_lib.CHAINS_F32)
    assert np.isfinite(losses).all()
    for b in range(B):
        g_o[b, xl[b]:] = 0.0
    U.assert_same(losses, l_o, 2e-6, 1e-6, "losses")
    U.assert_same(grads, g_o, F32_RTOL, 2e-5, "grads")
    assert np.abs(grads - g_o).max() > 0        # (not the f64 chains' result by accident)


@pytest.mark.parametrize("algo", ALGOS, ids=ALGO_IDS.get)
def test_wide_alphabet_sample_against_oracle(algo):
    # C5-like: V=8000, S<=64, T=256 on two utterances (per-utterance alphabet compaction around the lattice kernels)
    g = torch.Generator().manual_seed(5)
    B, T, V, S = 2, 256, 8000, 64
    x = torch.randn(B, T, V, generator=g)
    tg = torch.randint(1, V, (B, S), generator=g)
    tg[1, 3] = tg[1, 4]       # a repeat
    tl = torch.tensor([64, 40])
    xl = torch.tensor([256, 200])
    lp = torch.log_softmax(x.double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg.numpy(), xl.numpy(), tl.numpy(), 0)
    losses, grads = run(x, tg, xl, tl, 0, False, algo, l_o)
    for b in range(B):
        g_o[b, xl[b]:] = 0.0
    U.assert_same(losses, l_o, F32_RTOL, F32_ATOL, "losses")
    U.assert_same(grads, g_o, F32_RTOL, 5e-7, "grads")


def test_full_c2_properties():
    # full BASELINE size: size-independent properties (the oracle would take minutes here)
    g = torch.Generator().manual_seed(0)
    B, T, V, S = 256, 1000, 29, 200
    x = torch.randn(B, T, V, generator=g)
    tg = torch.randint(1, V, (B, S), generator=g)
    tl = torch.randint(S // 2, S + 1, (B,), generator=g)
    xl = torch.randint(T // 2, T + 1, (B,), generator=g)
    xl[:128] = T
    losses, grads = U.c_abi_loss(x, tg, xl, tl, 0, False)
    assert np.isfinite(losses).all() and (losses > 0).all()
    # rows of softmax - posterior sum to zero on valid frames and are exactly zero on padded frames
    rs = np.abs(grads.astype(np.float64).sum(-1))
    assert rs.max() < 5e-5
    for b in (0, 130, 200, 255):
        assert not grads[b, xl[b]:].any()
    # the posterior mass of the blank+labels is one per frame and never negative beyond rounding
    sm = torch.softmax(x.double(), -1).numpy()
    post = sm - grads
    for b in (0, 130, 255):
        n = int(xl[b])
        assert post[b, :n].min() > -1e-5
        np.testing.assert_allclose(post[b, :n].sum(-1), 1.0, atol=1e-4)
    # spot check eight utterances against the oracle
    idx = [0, 1, 127, 128, 129, 200, 254, 255]
    lp = torch.log_softmax(x[idx].double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg[idx].numpy(), xl[idx].numpy(), tl[idx].numpy(), 0)
    for k, b in enumerate(idx):
        g_o[k, xl[b]:] = 0.0
    U.assert_same(losses[idx], l_o, F32_RTOL, F32_ATOL, "losses")
    U.assert_same(grads[idx], g_o, F32_RTOL, F32_ATOL, "grads")


@pytest.mark.parametrize("algo", ALGOS, ids=ALGO_IDS.get)
def test_invalid_lengths_poison_not_crash(algo):
    x = torch.randn(3, 5, 4)
    losses, grads = U.c_abi_loss(x, [[1, 2], [1, 2], [1, 2]], [5, 0, 9], [2, 2, 2], 0, False, algo)
    assert np.isfinite(losses[0]) and np.isnan(losses[1]) and np.isnan(losses[2])
    assert np.isnan(grads[1]).all() and np.isnan(grads[2]).all() and np.isfinite(grads[0]).all()


@pytest.mark.parametrize("algo", ALGOS, ids=lambda a: ALGO_IDS[a])
@pytest.mark.parametrize("V", [4, 200])
def test_out_of_range_targets_poison_not_crash(algo, V):
    # a -1 padding value inside t_len, or an id == V: NaN for that utterance only (include/e2e_ctc.h), no memory fault
    if algo == _lib.ALGO_FAST and V > 96:
        pytest.skip("E2E_ALGO_FAST proper covers V <= 96; the wide path is reached through AUTO")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 9, V, generator=g)
    tg = [[1, 2, 3], [1, -1, 2], [1, V, 2], [1, V + 5, -7]]
    for dt in (torch.float32, torch.float64):
        if algo == _lib.ALGO_FAST and dt != torch.float32:
            continue
        losses, grads = U.c_abi_loss(x.to(dt), tg, [9, 9, 9, 9], [3, 3, 3, 1], 0, False, algo)
        assert np.isfinite(losses[0]) and np.isfinite(grads[0]).all()
        assert np.isnan(losses[1]) and np.isnan(losses[2]) and np.isnan(grads[1]).all() and np.isnan(grads[2]).all()
        assert np.isfinite(losses[3]) and np.isfinite(grads[3]).all()      # the bad entries lie beyond t_len


def test_scale_grads_in_place_both_dtypes():
    L = _lib.load()
    d = U.dev()
    for dt, code in ((torch.float32, _lib.F32), (torch.float64, _lib.F64)):
        g = torch.randn(5, 7, 3, dtype=dt, device=d)
        s = torch.tensor([1.0, 0.0, -2.5, 1.0, 3.0], dtype=dt, device=d)
        want = g * s.view(-1, 1, 1)
        _lib.check(L.e2e_ctc_scale_grads(g.data_ptr(), code, s.data_ptr(), 5, 21, _lib.stream_ptr(d)))
        torch.cuda.synchronize()
        assert torch.equal(g, want)
    assert L.e2e_ctc_scale_grads(None, 9, None, 1, 1, None) == -1


@pytest.mark.parametrize("algo", ALGOS, ids=lambda a: ALGO_IDS[a])
@pytest.mark.parametrize("shape", ["small", "wide", "f64"])
def test_options_grad_scale_and_fused_reduction(algo, shape):
    """e2e_ctc_loss_fwd_bwd_opt: the gradient comes out multiplied by grad_scale (NaN slabs stay NaN) and the sum / mean of
    the losses is written by the call itself -- on every path: fast (nothing flagged: the segment kernel's last wave;
    something flagged: the fallback's last workgroup), exact, wide."""
    g = torch.Generator().manual_seed(21)
    V = 300 if shape == "wide" else 7
    dt = torch.float64 if shape == "f64" else torch.float32
    if algo == _lib.ALGO_FAST and shape != "small":
        pytest.skip("E2E_ALGO_FAST proper: f32, V <= 96")
    B, T = 6, 40
    x = torch.randn(B, T, V, generator=g, dtype=dt)
    tg = torch.randint(1, V, (B, 9), generator=g)
    xl = [40, 31, 40, 12, 40, 25]
    tl = [9, 5, 0, 9, 3, 7]
    for feasible in (True, False):
        if not feasible:
            tl = [9, 5, 0, 9, 3, 7]
            xl = [40, 31, 40, 5, 40, 25]              # utterance 3: T < S -> +inf / NaN slab (flagged on the fast path)
        base_l, base_g = U.c_abi_loss(x, tg, xl, tl, 0, False, algo)
        for scale, red in ((0.125, _lib.REDUCE_MEAN), (-3.0, _lib.REDUCE_SUM), (1.0, _lib.REDUCE_NONE)):
            l2, g2, r = U.c_abi_loss(x, tg, xl, tl, 0, False, algo, opts=(scale, red))
            assert np.array_equal(l2, base_l, equal_nan=True)
            # (the wide path forms the label columns as softmax*s - (posterior part)*s: one more rounding of a difference)
            U.assert_same(g2, base_g * scale, 2e-5 if dt == torch.float32 else 1e-13, 1e-8 if dt == torch.float32 else 1e-30,
                          "scaled grads")
            if red == _lib.REDUCE_NONE:
                assert r == 7.0                       # untouched
            else:
                want = base_l.astype(np.float64).sum() / (B if red == _lib.REDUCE_MEAN else 1)
                if np.isfinite(want):
                    assert abs(r - want) <= (1e-6 if dt == torch.float32 else 1e-13) * abs(want)
                else:
                    assert (np.isnan(r) and np.isnan(want)) or r == want      # (FAST poisons what it flags: NaN)
    # log-prob mode: padded rows are exp(lp) * scale (quirk Q1 keeps its shape)
    lp = torch.log_softmax(x, -1)
    l0, g0 = U.c_abi_loss(lp, tg, xl, tl, 0, True, algo)
    l1, g1, _ = U.c_abi_loss(lp, tg, xl, tl, 0, True, algo, opts=(0.5, _lib.REDUCE_NONE))
    U.assert_same(g1, g0 * 0.5, 2e-5 if dt == torch.float32 else 1e-13, 1e-8, "scaled grads, log-prob mode")


def test_argument_errors_are_reported():
    L = _lib.load()
    rc = L.e2e_ctc_loss_fwd_bwd(None, 5, 1, 1, 1, 1, None, 0, None, None, 1, 1, 1, 0, 0, None, None, None, 0, 0, None)
    assert rc == -1 and b"dtype" in L.e2e_last_error()


def test_fast_path_falls_back_on_blank_valued_targets_and_tiny_probabilities():
    # (a) a target equal to the blank id shares the blank column in the reference (ctc_loss.cpp:53,109-113);
    # (b) log-probs so peaked that the scaled f32 segment rows underflow.  AUTO must still match the oracle.
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 30, 6, generator=g)
    tg = torch.tensor([[1, 0, 2, 3], [1, 2, 3, 4]])
    lp = torch.log_softmax(x.double(), -1)
    lp[1, 5:25, 1:5] -= 80.0          # twenty frames where every label is ~e-80: forces the lattice out of range
    lp = lp.float()
    xl, tl = [30, 30], [4, 4]
    l_o, g_o = O.ctc_loss(lp.double().numpy(), tg.numpy(), xl, tl, 0)
    losses, grads = U.c_abi_loss(lp, tg, xl, tl, 0, True, _lib.ALGO_AUTO)
    U.assert_same(losses, l_o, F32_RTOL, F32_ATOL, "losses")
    U.assert_same(grads, g_o, F32_RTOL, F32_ATOL, "grads")
    lf, _ = U.c_abi_loss(lp, tg, xl, tl, 0, True, _lib.ALGO_FAST)
    assert np.isnan(lf[0])            # (a) is always handed to the exact kernel


@pytest.mark.parametrize("shape", [(321, 57, 250, 1.0), (195, 84, 150, 0.1), (418, 24, 220, 3.0), (230, 40, 215, 0.1)],
                         ids=lambda s: "T%d_V%d_S%d" % s[:3])
def test_fast_path_dense_targets(shape):
    """Targets almost as long as the input (S/T 0.5 .. 0.93): the band of reachable cells is narrow, most lanes of the
    first checkpoint row are exactly zero and the tilt is at its clip.  (A randomised sweep, tools/diag/
    fuzz_fast_vs_exact.py, found gradients off by up to 0.7 in frames 16..31 of such utterances, unflagged: all-zero lanes
    had slid 24 bits per lane down in their exponent unit, which pushed the beta rows of those lanes out of f32 range.)"""
    T, V, S, sharp = shape
    g = torch.Generator().manual_seed(T + V)
    x = torch.randn(3, T, V, generator=g, dtype=torch.float64) * sharp
    lp = torch.log_softmax(x, -1)
    tg = torch.randint(1, V, (3, S), generator=g)
    xl = torch.tensor([T, T - 7, (2 * T) // 3])
    tl = torch.tensor([S, S - 20, min(S, (2 * T) // 3 - 30)])
    l_o, g_o = O.ctc_loss(lp.numpy(), tg.numpy(), xl.numpy(), tl.numpy(), 0)
    losses, grads = run(lp.float(), tg, xl, tl, 0, True, _lib.ALGO_FAST, np.where(np.isfinite(l_o), l_o, np.nan))
    feasible = np.isfinite(l_o)
    assert feasible.any()
    U.assert_same(losses[feasible], l_o[feasible], F32_RTOL, F32_ATOL, "losses")
    U.assert_same(grads[feasible], g_o[feasible], F32_RTOL, F32_ATOL, "grads")


def test_range_flags_are_redone_in_f64_not_by_the_exact_kernel():
    """Logits at scale 8 that have nothing to do with the targets (loss in the thousands): rows of the lattice span more than
    f32 holds and the f32 segment kernel leaves its range or fails its partition-sum self-check for most utterances (flag
    bits 8 / 16).  ALGO_AUTO then redoes only the segments in f64 from the chains' checkpoints (the full log-domain
    recomputation only if a row fails to reproduce the chains' log Z) -- the result must be the exact kernel's."""
    rng = np.random.default_rng(0)
    B, T, V, S = 24, 700, 29, 150
    x = (rng.standard_normal((B, T, V)) * 8.0).astype(np.float32)
    tg = rng.integers(1, V, size=(B, S)); tl = rng.integers(S // 2, S + 1, size=B); xl = np.full(B, T); xl[1:] -= rng.integers(0, 60, size=B - 1)
    args = (torch.from_numpy(x), torch.from_numpy(tg), torch.from_numpy(xl), torch.from_numpy(tl), 0, False)
    lf, _ = U.c_abi_loss(*args, _lib.ALGO_FAST)
    la, ga = U.c_abi_loss(*args, _lib.ALGO_AUTO)
    le, ge = U.c_abi_loss(*args, _lib.ALGO_EXACT)
    assert np.isnan(lf).sum() >= 2, "this input no longer drives the f32 segment kernel out of range: pick a harder one"
    U.assert_same(la, le, F32_RTOL, 2e-5, "losses")
    U.assert_same(ga, ge, F32_RTOL, F32_ATOL, "grads")


@pytest.mark.parametrize("shape", [(16, 900, 29, 300), (8, 600, 448, 100)], ids=lambda s: "B%d_T%d_V%d_S%d" % s)
def test_range_flags_with_eight_pairs_per_lane_are_redone_in_f64_too(shape):
    """The same regime where the segment kernel holds eight label pairs per lane -- targets of 256..447 labels, alphabets
    beyond 224 columns: until round 4 those utterances were recomputed in full by the exact kernel (8.5 ms instead of 0.27 for
    a headline-sized batch with 88 of them); the flagged launch now has an instance whose f64 redo of single segments takes
    16 cells per lane.  Most utterances must be settled by that redo (few are left for the full recomputation), and the
    result must be the exact kernel's."""
    B, T, V, S = shape
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((B, T, V)) * 3.0).astype(np.float32)
    tg = rng.integers(1, V, size=(B, S)); tl = rng.integers(max(1, (2 * S) // 3), S + 1, size=B); xl = np.full(B, T); xl[1:] -= rng.integers(0, 60, size=B - 1)
    args = (torch.from_numpy(x), torch.from_numpy(tg), torch.from_numpy(xl), torch.from_numpy(tl), 0, False)
    lf, _ = U.c_abi_loss(*args, _lib.ALGO_FAST)
    keep = {}
    la, ga = U.c_abi_loss(*args, _lib.ALGO_AUTO, keep=keep)
    le, ge = U.c_abi_loss(*args, _lib.ALGO_EXACT)
    nflag = int(np.isnan(lf).sum())
    assert nflag >= 2, "this input no longer drives the f32 segment kernel out of range: pick a harder one"
    import ctypes
    cnt = ctypes.c_int(-1)
    L = _lib.load()
    L.e2e_debug_fast_redo_failures.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    assert L.e2e_debug_fast_redo_failures(keep["workspace"].data_ptr(), B, T, V, S, ctypes.byref(cnt)) == 0
    assert 0 <= cnt.value <= nflag // 2, "%d of %d flagged utterances were left to the full recomputation" % (cnt.value, nflag)
    U.assert_same(la, le, F32_RTOL, 2e-5, "losses")
    U.assert_same(ga, ge, F32_RTOL, F32_ATOL, "grads")


def test_fast_path_takes_peaky_consistent_emissions():
    """What a trained model emits: logits that favour a valid alignment of the utterance's own targets by 2 .. 20 over unit
    noise.  The fast path must handle these itself (no flag, no exact fallback) and accurately."""
    rng = np.random.default_rng(5)
    B, T, V, S = 6, 240, 29, 48
    for boost in (2.0, 6.0, 20.0):
        x = rng.standard_normal((B, T, V))
        tg = rng.integers(1, V, size=(B, S)); tl = rng.integers(S // 2, S + 1, size=B); xl = np.array([T, T, T - 9, T - 40, T, 200])
        for b in range(B):
            L = int(tl[b])
            slots = np.sort(rng.choice(np.arange(0, int(xl[b]), 2), size=L, replace=False))     # never adjacent
            x[b, slots, tg[b, :L]] += boost
            rest = np.setdiff1d(np.arange(int(xl[b])), slots)
            x[b, rest, 0] += boost
        lp = torch.log_softmax(torch.from_numpy(x), -1)
        l_o, g_o = O.ctc_loss(lp.numpy(), tg, xl, tl, 0)
        losses, grads = U.c_abi_loss(lp.float(), torch.from_numpy(tg), torch.from_numpy(xl), torch.from_numpy(tl), 0, True, _lib.ALGO_FAST)
        assert not np.isnan(losses).any(), "boost %g: the fast path gave up on %s" % (boost, np.nonzero(np.isnan(losses))[0].tolist())
        U.assert_same(losses, l_o, F32_RTOL, 2e-5, "losses")
        U.assert_same(grads, g_o, F32_RTOL, F32_ATOL, "grads")


@pytest.mark.parametrize("algo", ALGOS, ids=ALGO_IDS.get)
@pytest.mark.parametrize("logprobs", [False, True])
def test_wide_alphabet_edge_cases(algo, logprobs):
    # V=203 (not a multiple of 4: scalar row loops), ragged lengths, repeats, an empty target, an infeasible utterance,
    # blank in the middle of the alphabet, time-major strides
    g = torch.Generator().manual_seed(77)
    B, T, V, S = 5, 40, 203, 12
    x = torch.randn(T, B, V, generator=g).permute(1, 0, 2)          # (B,T,V) view of a time-major tensor
    if logprobs:
        x = torch.log_softmax(x, -1)
    tg = torch.randint(0, V - 1, (B, S), generator=g)
    tg[tg == 100] = 101                                              # blank id 100 must not appear
    tg[1, 3] = tg[1, 4] = tg[1, 5]                                   # repeats
    xl = torch.tensor([40, 31, 40, 9, 22])
    tl = torch.tensor([12, 7, 0, 12, 5])                             # utterance 3: T=9 < S=12 -> infeasible
    lp = (x if logprobs else torch.log_softmax(x.double(), -1)).double().numpy()
    l_o, g_o = O.ctc_loss(lp, tg.numpy(), xl.numpy(), tl.numpy(), 100)
    if not logprobs:
        for b in range(B):
            if np.isfinite(l_o[b]):
                g_o[b, xl[b]:] = 0.0
    losses, grads = run(x, tg, xl, tl, 100, logprobs, algo, l_o)
    U.assert_same(losses, l_o, F32_RTOL, F32_ATOL, "losses")
    U.assert_same(grads, g_o, F32_RTOL, 1e-6, "grads")


@pytest.mark.parametrize("algo", ALGOS, ids=ALGO_IDS.get)
@pytest.mark.parametrize("logprobs", [False, True])
@pytest.mark.parametrize("V", [204, 2052, 4100])
def test_wide_alphabet_edge_cases_single_read(algo, logprobs, V):
    # contiguous rows of a multiple of 4 columns take the single-read form (the row is held in a wave's registers:
    # 8 / 16 / 32 float4 per lane for these widths, the last group partial); same corner cases as above -- ragged
    # lengths (padded frames: exp(lp) in log-prob mode, 0 for fused logits), repeats, an empty target, an infeasible
    # utterance whose slab is poisoned after the lattice, blank in the middle of the alphabet
    g = torch.Generator().manual_seed(78 + V)
    B, T, S = 5, 40, 12
    x = torch.randn(B, T, V, generator=g) * 2
    if logprobs:
        x = torch.log_softmax(x, -1)
    tg = torch.randint(0, V - 1, (B, S), generator=g)
    tg[tg == 100] = 101                                              # blank id 100 must not appear
    tg[1, 3] = tg[1, 4] = tg[1, 5]                                   # repeats
    xl = torch.tensor([40, 31, 40, 9, 22])
    tl = torch.tensor([12, 7, 0, 12, 5])                             # utterance 3: T=9 < S=12 -> infeasible
    lp = (x if logprobs else torch.log_softmax(x.double(), -1)).double().numpy()
    l_o, g_o = O.ctc_loss(lp, tg.numpy(), xl.numpy(), tl.numpy(), 100)
    if not logprobs:
        for b in range(B):
            if np.isfinite(l_o[b]):
                g_o[b, xl[b]:] = 0.0
    losses, grads = run(x, tg, xl, tl, 100, logprobs, algo, l_o)
    U.assert_same(losses, l_o, F32_RTOL, F32_ATOL, "losses")
    U.assert_same(grads, g_o, F32_RTOL, 1e-6, "grads")


def test_full_c5_shape_properties():
    # one GPU's share of BASELINE configs[4]: B=512, T=256, V=8000, S<=64 (4.2 GB of logits)
    g = torch.Generator().manual_seed(5)
    B, T, V, S = 512, 256, 8000, 64
    x = torch.randn(B, T, V, generator=g, dtype=torch.float32)
    tg = torch.randint(1, V, (B, S), generator=g)
    tl = torch.randint(S // 2, S + 1, (B,), generator=g)
    xl = torch.randint(200, T + 1, (B,), generator=g)
    d = U.dev()
    xd = x.to(d)
    losses, grads = U.c_abi_loss(xd, tg, xl, tl, 0, False)
    assert np.isfinite(losses).all() and (losses > 0).all()
    idx = [0, 255, 511]
    sub = grads[idx].astype(np.float64)
    assert np.abs(sub.sum(-1)).max() < 2e-4                      # rows of softmax - posterior sum to zero
    for k, b in enumerate(idx):
        assert not sub[k, xl[b]:].any()
    lp = torch.log_softmax(x[idx].double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg[idx].numpy(), xl[idx].numpy(), tl[idx].numpy(), 0)
    for k, b in enumerate(idx):
        g_o[k, xl[b]:] = 0.0
    U.assert_same(losses[idx], l_o, F32_RTOL, F32_ATOL, "losses")
    U.assert_same(grads[idx], g_o, F32_RTOL, 5e-7, "grads")


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
def test_word_piece_vocabularies_are_served_whatever_the_dtype(dtype):
    """V = 32 000 with more than 95 distinct labels (and f64 input at any target length): the exact kernel keeps one BIT per
    alphabet column in LDS, not a double -- no (V, S, dtype) is refused (src/losses/ctc_loss.cpp:25-36 has no bounds)."""
    g = torch.Generator().manual_seed(9)
    B, T, V, S = 2, 150, 32000, 120
    x = torch.randn(B, T, V, generator=g).to(dtype)
    tg = torch.randint(1, V, (B, S), generator=g)
    tg[0, 5] = tg[0, 4]; tg[1, 7] = 0                        # a repeat, and a target equal to the blank id
    xl = torch.tensor([T, T - 13]); tl = torch.tensor([S, 70])
    losses, grads = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_AUTO)
    lp = torch.log_softmax(x.double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg.numpy(), xl.numpy(), tl.numpy(), 0)
    g_o[1, T - 13:] = 0
    rt, at = (F32_RTOL, F32_ATOL) if dtype == torch.float32 else (1e-9, 1e-12)
    U.assert_same(losses, l_o, rt, at * 10, "losses")
    U.assert_same(grads, g_o, rt, at, "grads")


@pytest.mark.parametrize("shape", [(2, 400, 500, 150), (2, 700, 29, 300), (1, 650, 3000, 300), (2, 300, 97, 120),
                                   (2, 256, 8000, 200), (1, 2000, 29, 400), (2, 1300, 29, 500), (1, 1100, 40, 520)],
                         ids=lambda s: "B%d_T%d_V%d_S%d" % s)
def test_shapes_outside_the_fast_paths_are_still_served(shape):
    """Targets longer than the fast kernels take (447 labels), or more than 95 distinct labels at an alphabet beyond 96
    columns: the exact kernel must serve these (no E2E_ERR_UNSUPPORTED, the reference has no such bounds) -- under AUTO
    with f32 in its scaled probability-domain form (two cells per thread: up to 511 labels), beyond that in the log domain."""
    B, T, V, S = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, T, V, generator=g)
    tg = torch.randint(1, V, (B, S), generator=g)
    xl = torch.full((B,), T)
    tl = torch.tensor([S] + [S // 2] * (B - 1))
    losses, grads = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_AUTO)
    lp = torch.log_softmax(x.double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg.numpy(), xl.numpy(), tl.numpy(), 0)
    U.assert_same(losses, l_o, F32_RTOL, F32_ATOL, "losses")
    U.assert_same(grads, g_o, F32_RTOL, F32_ATOL, "grads")


def test_more_fully_recomputed_utterances_than_alpha_slabs_and_a_mixed_batch():
    """The flagged-utterance launch keeps 24 alpha slabs: 60 utterances that all need the reference's arithmetic (infeasible
    alignments, a blank-valued target) are strided over them; range-flagged ones in the same batch have their segments
    redone in f64 by the launch's other waves; the fused mean must be that of the final losses."""
    rng = np.random.default_rng(11)
    B, T, V, S = 96, 260, 29, 140
    x = (rng.standard_normal((B, T, V))).astype(np.float32)
    x[60:80] *= 8.0                                             # sharp, unrelated: the f32 segment rows leave their range
    tg = rng.integers(1, V, size=(B, S)); tl = rng.integers(60, 100, size=B); xl = np.full(B, T)
    xl[:50] = rng.integers(20, 56, size=50)                     # fewer frames than labels (>= 60): infeasible
    tg[50:60, 3] = 0                                            # a target equal to the blank id
    xt = torch.from_numpy(x)
    lp = torch.log_softmax(xt.double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg, xl, tl, 0)
    for b in range(50, B):
        g_o[b, xl[b]:] = 0                                      # fused logits: padded frames get 0
    g_o[:50] = np.nan                                           # (quirk Q2: an infeasible utterance's whole slab)
    la, ga, red = U.c_abi_loss(xt, tg, xl, tl, 0, False, _lib.ALGO_AUTO, opts=(1.0, _lib.REDUCE_SUM))
    assert np.isinf(l_o[:50]).all() and np.isfinite(l_o[50:]).all()
    U.assert_same(la, l_o, F32_RTOL, 2e-5, "losses")
    U.assert_same(ga, g_o, F32_RTOL, F32_ATOL, "grads")
    assert np.isinf(red) and red > 0
    la2, _, red2 = U.c_abi_loss(xt[50:], tg[50:], xl[50:], tl[50:], 0, False, _lib.ALGO_AUTO, opts=(1.0, _lib.REDUCE_MEAN))
    assert abs(red2 - l_o[50:].mean()) <= 1e-5 * abs(l_o[50:].mean())


@pytest.mark.parametrize("V", [2, 300])
@pytest.mark.parametrize("algo", [_lib.ALGO_AUTO, _lib.ALGO_EXACT], ids=ALGO_IDS.get)
def test_no_alignment_and_a_blank_valued_target_leaves_the_references_element_pattern(algo, V):
    """Invalid input twice over: the reference's alpha refuses the skip into a label equal to the blank id, its beta takes
    it, so alpha + beta can be finite where log Z is -inf -- exp(log_post - logZ) then leaves -inf in those columns and NaN
    in the others (ctc_loss.cpp:102-117).  Found by tools/diag/fuzz_scaled_vs_oracle.py; utterances 1..3 are ordinary;
    V = 300 goes through the wide-alphabet path, which has to carry the pattern out of its compact columns."""
    x = torch.randn(4, 2, V, generator=torch.Generator().manual_seed(640)) * 8.0
    lp = torch.log_softmax(x.double(), -1)
    tg = np.array([[1, 0], [0, 1], [1, 1], [0, 1]]); xl = [2, 1, 1, 2]; tl = [2, 1, 0, 1]
    l_o, g_o = O.ctc_loss(lp.numpy(), tg, xl, tl, 0)
    assert np.isinf(l_o[0]) and np.isneginf(g_o[0]).any() and np.isnan(g_o[0]).any() and np.isfinite(l_o[1:]).all()
    losses, grads = U.c_abi_loss(lp.float(), tg, xl, tl, 0, True, algo)
    U.assert_same(losses, l_o, F32_RTOL, F32_ATOL, "losses")
    U.assert_same(grads, g_o, F32_RTOL, F32_ATOL, "grads")


def test_scaled_exact_form_hands_over_to_the_log_domain_beyond_f64_range():
    """Under AUTO the exact kernel walks f32 utterances in the probability domain (f64 rows rescaled by powers of two).
    Log-probabilities below f64's exponent range (exp(-800) == 0) leave that form without a partition sum: it must hand the
    utterance to the log-domain walk, whose answer the oracle gives; a row that merely drives the *product* out of range
    (-300 a frame) stays in the scaled form.  Both utterances reach the exact kernel through a blank-valued label."""
    T, V = 12, 6
    lp = np.full((2, T, V), -800.0, dtype=np.float32)
    lp[1] = -300.0
    rng = np.random.default_rng(3)
    lp += rng.uniform(-2, 0, size=lp.shape).astype(np.float32)
    tg = np.array([[1, 0, 3], [2, 0, 2]]); xl = [T, T - 2]; tl = [3, 3]
    l_o, g_o = O.ctc_loss(lp.astype(np.float64), tg, xl, tl, 0)
    assert np.isfinite(l_o).all() and l_o[0] > 9000 and l_o[1] > 2900
    losses, grads = U.c_abi_loss(torch.from_numpy(lp), tg, xl, tl, 0, True, _lib.ALGO_AUTO)
    U.assert_same(losses, l_o, F32_RTOL, F32_ATOL, "losses")
    U.assert_same(grads, g_o, F32_RTOL, F32_ATOL, "grads")
    le, ge = U.c_abi_loss(torch.from_numpy(lp), tg, xl, tl, 0, True, _lib.ALGO_EXACT)
    U.assert_same(le, losses, 1e-6, 0, "losses: scaled / handed-over against the log-domain kernel")
    U.assert_same(ge, grads, 1e-6, 1e-9, "grads: scaled / handed-over against the log-domain kernel")


def test_scaled_exact_form_scales_rows_by_their_feasible_cells():
    """A transcript barely shorter than the utterance with the blank dominating every frame: the all-blank path at j = 0 is
    doomed (outside the reference's [start, end) window almost from the beginning) and ~2^500 above everything feasible.  The
    scaled form takes each row's power of two from the cells inside the window, so the feasible cells keep their bits; both
    utterances reach the exact kernel through a blank-valued label (AUTO) and are compared with the oracle and with the
    log-domain kernel."""
    B, T, V, S = 2, 96, 8, 90
    rng = np.random.default_rng(11)
    x = rng.normal(0, 0.3, size=(B, T, V)).astype(np.float32)
    x[:, :, 0] += 6.0                                              # blank probability ~0.98, labels ~0.003
    tg = rng.integers(1, V, size=(B, S)); tg[:, 1::2] = 1; tg[:, 0::2] = 2          # no repeats: T - S = 6 spare frames
    tg[0, 40] = 0                                                  # a blank-valued label: the fast path hands the utterance over
    tg[1, 10] = 0
    xl = [T, T - 2]; tl = [S, S - 3]
    lp = torch.log_softmax(torch.from_numpy(x).double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg, xl, tl, 0)
    assert np.isfinite(l_o).all() and (l_o > 400).all()
    losses, grads = U.c_abi_loss(torch.from_numpy(lp.astype(np.float32)), tg, xl, tl, 0, True, _lib.ALGO_AUTO)
    l32, g32 = O.ctc_loss(lp.astype(np.float32).astype(np.float64), tg, xl, tl, 0)
    U.assert_same(losses, l32, F32_RTOL, F32_ATOL, "losses")
    U.assert_same(grads, g32, F32_RTOL, F32_ATOL, "grads")
    le, ge = U.c_abi_loss(torch.from_numpy(lp.astype(np.float32)), tg, xl, tl, 0, True, _lib.ALGO_EXACT)
    U.assert_same(le, losses, 1e-6, 0, "losses: scaled form against the log-domain kernel")
    U.assert_same(ge, grads, 1e-6, 1e-9, "grads: scaled form against the log-domain kernel")


@pytest.mark.parametrize("shape", [(3, 700, 29, 300), (2, 2000, 29, 400), (2, 1100, 48, 447), (2, 1000, 96, 425), (2, 900, 80, 300)],
                         ids=lambda s: "B%d_T%d_V%d_S%d" % s)
def test_long_transcripts_take_the_fast_path(shape):
    """Targets of 256..447 labels: the halo chains on four waves per direction and the segment kernel with
    eight pairs per lane -- served by the fast path itself (ALGO_FAST leaves no NaN), equal to the oracle at the default
    tolerances, ragged lengths included."""
    B, T, V, S = shape
    g = torch.Generator().manual_seed(15)
    x = torch.randn(B, T, V, generator=g)
    tg = torch.randint(1, V, (B, S), generator=g)
    xl = torch.tensor([T] + [T - 37 * (b + 1) for b in range(B - 1)])
    tl = torch.tensor([S] + [257 + 11 * b for b in range(B - 1)])
    lf, gf = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_FAST)
    assert np.isfinite(lf).all(), "the fast path flagged %d of %d utterances" % (int(np.isnan(lf).sum()), B)
    la, ga = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_AUTO)
    lp = torch.log_softmax(x.double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg.numpy(), xl.numpy(), tl.numpy(), 0)
    for b in range(B):
        g_o[b, xl[b]:] = 0
    for losses, grads in ((lf, gf), (la, ga)):
        U.assert_same(losses, l_o, F32_RTOL, F32_ATOL * 100, "losses")
        U.assert_same(grads, g_o, F32_RTOL, F32_ATOL, "grads")


@pytest.mark.parametrize("logprobs", [False, True], ids=["logits", "logprobs"])
@pytest.mark.parametrize("shape", [(2, 300, 97, 120), (3, 200, 128, 100), (2, 256, 129, 150), (3, 400, 224, 223), (2, 300, 200, 40),
                                   (4, 64, 177, 30)], ids=lambda s: "B%d_T%d_V%d_S%d" % s)
def test_alphabets_of_97_to_224_columns_take_the_fast_path(shape, logprobs):
    """alphabets beyond 96 columns -- above all the compacted word-piece targets of more than 95 pieces --
    run on the fast lattice kernels (ChainF64W: halo chains over an f32 probability ring filled from a table that
    ctc_fast_prob_kernel computes once; the segment kernel's wide-row form): ALGO_FAST leaves no NaN, results equal the
    oracle at the default tolerances; ragged lengths, a repeated label.  (src/losses/ctc_loss.cpp:25-36: no bound on V.)"""
    B, T, V, S = shape
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, T, V, generator=g)
    if logprobs:
        x = torch.log_softmax(x, -1)
    tg = torch.randint(1, V, (B, S), generator=g)
    tg[0, 3] = tg[0, 2]
    xl = torch.tensor([T] + [T - 9 * (b + 1) for b in range(B - 1)])
    tl = torch.tensor([S] + [max(1, S - 17 * (b + 1)) for b in range(B - 1)])
    lf, gf = U.c_abi_loss(x, tg, xl, tl, 0, logprobs, _lib.ALGO_FAST)
    assert np.isfinite(lf).all(), "the fast path flagged %d of %d utterances" % (int(np.isnan(lf).sum()), B)
    lp = (x.double() if logprobs else torch.log_softmax(x.double(), -1)).numpy()
    l_o, g_o = O.ctc_loss(lp, tg.numpy(), xl.numpy(), tl.numpy(), 0)
    if not logprobs:
        for b in range(B):
            g_o[b, xl[b]:] = 0
    U.assert_same(lf, l_o, F32_RTOL, F32_ATOL * 10, "losses")
    U.assert_same(gf, g_o, F32_RTOL, F32_ATOL, "grads")


@pytest.mark.parametrize("shape", [(2, 256, 8000, 200), (2, 150, 32000, 120), (3, 300, 4096, 223), (3, 298, 230, 223),
                                   (2, 178, 3001, 167), (1, 260, 9001, 200)], ids=lambda s: "B%d_T%d_V%d_S%d" % s)
def test_word_piece_targets_of_more_than_95_pieces_take_the_fast_lattice(shape):
    """The wide path's compaction leaves up to Smax + 1 columns; beyond 96 of them the lattice used to be the exact kernel's
    (0.73 ms at B=64, T=256, V=8000, S<=200).  Under ALGO_FAST -- no fallback -- nothing may come back NaN-poisoned."""
    B, T, V, S = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, T, V, generator=g)
    tg = torch.randint(1, V, (B, S), generator=g)
    tg[0, 5] = tg[0, 4]
    xl = torch.tensor([T] + [T - 13] * (B - 1)); tl = torch.tensor([S] + [S - 30] * (B - 1))
    lf, gf = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_FAST)
    assert np.isfinite(lf).all(), "the fast path flagged %d of %d utterances" % (int(np.isnan(lf).sum()), B)
    lp = torch.log_softmax(x.double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg.numpy(), xl.numpy(), tl.numpy(), 0)
    for b in range(B):
        g_o[b, xl[b]:] = 0
    U.assert_same(lf, l_o, F32_RTOL, F32_ATOL * 10, "losses")
    U.assert_same(gf, g_o, F32_RTOL, 5e-7, "grads")


@pytest.mark.parametrize("shape", [(2, 500, 225, 224), (2, 700, 300, 300), (2, 900, 448, 447), (2, 300, 97, 250), (2, 700, 8000, 300),
                                   (1, 600, 9001, 447)], ids=lambda s: "B%d_T%d_V%d_S%d" % s)
def test_up_to_448_columns_and_447_labels_take_the_fast_path(shape):
    """More than 96 columns together with targets of 224..447 labels, or 225..448 columns with any target: the long-transcript
    kernels' wide-row form (ChainF64LW: f32 ring two blocks deep, 56 columns per producer lane; the segment kernel with eight
    pairs per lane and up to seven labels per gradient lane) -- directly and behind the wide path's compaction.  ALGO_FAST
    leaves no NaN; the oracle's results at the default tolerances."""
    B, T, V, S = shape
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, T, V, generator=g)
    tg = torch.randint(1, V, (B, S), generator=g)
    tg[0, 3] = tg[0, 2]
    xl = torch.tensor([T] + [T - 21 * (b + 1) for b in range(B - 1)])
    tl = torch.tensor([S] + [max(1, S - 50 * (b + 1)) for b in range(B - 1)])
    lf, gf = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_FAST)
    assert np.isfinite(lf).all(), "the fast path flagged %d of %d utterances" % (int(np.isnan(lf).sum()), B)
    lp = torch.log_softmax(x.double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg.numpy(), xl.numpy(), tl.numpy(), 0)
    for b in range(B):
        g_o[b, xl[b]:] = 0
    U.assert_same(lf, l_o, F32_RTOL, F32_ATOL * 100, "losses")
    U.assert_same(gf, g_o, F32_RTOL, F32_ATOL, "grads")


def test_wide_row_form_takes_strided_input_and_any_blank():
    """ChainF64W's probability table is filled through the logits' strides and the blank may sit anywhere: a time-major view
    (the reference's own layout before its transpose, modules/ctc_loss.py:33-36) with the blank in the last column and in
    the middle, against the oracle."""
    g = torch.Generator().manual_seed(23)
    B, T, V, S = 3, 180, 170, 120
    xt = torch.randn(T, B, V, generator=g)                  # time-major storage
    x = xt.permute(1, 0, 2)                                  # (B, T, V) view, strides (V, B*V, 1)
    assert not x.is_contiguous()
    for blank in (V - 1, 77):
        labs = [v for v in range(V) if v != blank]
        tg = torch.tensor(np.random.default_rng(blank).choice(labs, size=(B, S)), dtype=torch.long)
        xl = torch.tensor([T, T - 11, T - 40]); tl = torch.tensor([S, S - 13, 50])
        lf, gf = U.c_abi_loss(x, tg, xl, tl, blank, False, _lib.ALGO_FAST)
        assert np.isfinite(lf).all()
        lp = torch.log_softmax(x.double(), -1).numpy()
        l_o, g_o = O.ctc_loss(lp, tg.numpy(), xl.numpy(), tl.numpy(), blank)
        for b in range(B):
            g_o[b, xl[b]:] = 0
        U.assert_same(lf, l_o, F32_RTOL, F32_ATOL * 10, "losses (blank %d)" % blank)
        U.assert_same(gf, g_o, F32_RTOL, F32_ATOL, "grads (blank %d)" % blank)


def test_tiny_probabilities_at_a_wide_alphabet_are_handed_to_the_exact_kernel():
    """ChainF64W's probability table marks a FINITE log-probability below -69 (f32 would flush what the lattice makes of it);
    the producers flag the utterance (reason bit 64) and the exact kernel recomputes it: AUTO equals the oracle, FAST poisons
    that utterance only.  A log-probability of -inf -- an impossible symbol -- is exact and flags nothing."""
    g = torch.Generator().manual_seed(21)
    B, T, V, S = 3, 60, 150, 20
    lp = torch.log_softmax(torch.randn(B, T, V, generator=g, dtype=torch.float64), -1)
    tg = torch.randint(1, V, (B, S), generator=g)
    lp[1, 10:20, 1:] -= 85.0                     # ten frames where every label is ~e-90
    lp[2, :, 149] = -float("inf")               # a symbol that cannot occur ...
    tg[2][tg[2] == 149] = 7                      # ... and does not
    lp = lp.float()
    xl, tl = [T] * B, [S] * B
    l_o, g_o = O.ctc_loss(lp.double().numpy(), tg.numpy(), xl, tl, 0)
    losses, grads = U.c_abi_loss(lp, tg, xl, tl, 0, True, _lib.ALGO_AUTO)
    U.assert_same(losses, l_o, F32_RTOL, F32_ATOL, "losses")
    U.assert_same(grads, g_o, F32_RTOL, F32_ATOL, "grads")
    lf, gf = U.c_abi_loss(lp, tg, xl, tl, 0, True, _lib.ALGO_FAST)
    assert np.isnan(lf[1]) and np.isfinite(lf[[0, 2]]).all()
    U.assert_same(gf[[0, 2]], g_o[[0, 2]], F32_RTOL, F32_ATOL, "grads of the utterances the fast path kept")


def test_flagged_launch_with_more_utterances_than_its_flag_cache():
    """3000 utterances (the flagged launch caches 2048 flag words in LDS and reads the rest from memory; 24 alpha slabs serve
    every utterance that needs the reference's arithmetic): blank-valued targets and impossible lengths sprinkled over the
    whole batch, the sum of the losses written by the same call."""
    rng = np.random.default_rng(21)
    B, T, V, S = 3000, 48, 20, 12
    x = rng.standard_normal((B, T, V)).astype(np.float32)
    tg = rng.integers(1, V, size=(B, S)); tl = rng.integers(4, S + 1, size=B); xl = rng.integers(30, T + 1, size=B)
    hard = rng.choice(B, size=90, replace=False)
    tg[hard[:45], 2] = 0                                        # a target equal to the blank id
    xl[hard[45:]] = 3                                           # fewer frames than labels
    xt = torch.from_numpy(x)
    lp = torch.log_softmax(xt.double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg, xl, tl, 0)
    feasible = np.isfinite(l_o)
    for b in range(B):
        if feasible[b]:
            g_o[b, xl[b]:] = 0
    la, ga, red = U.c_abi_loss(xt, tg, xl, tl, 0, False, _lib.ALGO_AUTO, opts=(1.0, _lib.REDUCE_SUM))
    assert (~feasible).sum() >= 40 and (~feasible[2048:]).sum() >= 5
    U.assert_same(la, l_o, F32_RTOL, 2e-5, "losses")
    U.assert_same(ga[feasible], g_o[feasible], F32_RTOL, F32_ATOL, "grads")
    assert np.isnan(ga[~feasible]).all() and np.isinf(red)
    ok = np.flatnonzero(feasible)
    la2, _, red2 = U.c_abi_loss(xt[ok], tg[ok], xl[ok], tl[ok], 0, False, _lib.ALGO_AUTO, opts=(1.0, _lib.REDUCE_SUM))
    assert abs(red2 - l_o[ok].sum()) <= 2e-6 * abs(l_o[ok].sum())


def test_bandwidth_probe_copies_exactly():
    """`e2e_debug_stream_copy` (include/e2e_ctc_debug.h; what bench.py's `roofline.peak_measured` times): whole 32 KB pieces per
    wave plus a remainder -- the destination must equal the source for sizes on both sides of a piece."""
    L = _lib.load()
    d = U.dev()
    for n in (4 * 5, 1 << 13, (1 << 20) + 4 * 777):
        src = torch.randn(n, device=d)
        dst = torch.zeros_like(src)
        _lib.check(L.e2e_debug_stream_copy(dst.data_ptr(), src.data_ptr(), n * 4, _lib.stream_ptr(d)))
        torch.cuda.synchronize()
        assert torch.equal(src, dst), n
