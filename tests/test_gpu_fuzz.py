"""-m gpu: fixed-seed slices of the randomised sweeps (tools/diag/fuzz_*_vs_oracle.py, fuzz_align.py) in the driver-run suite.

Every case is compared with the ORACLE (oracle/ctc_oracle.c through tests/oracle_lib.py), never with another HIP kernel.
Round 5's three latent bugs (an f64 segment redo that wrote NaN rows and reported success, a call that failed for 73..96
columns with more than 255 labels, a wave vote inside `if (lane == 0)`) were all found by these generators while the suite,
which only had hand-picked shapes, stayed green for two rounds -- hence this file.

Families (cases; what they drive):
  ext      300 f32 + 70 bf16 + 70 f16   E2E_ALGO_AUTO on emissions that leave the f32 lattice: the f64 segment redo, the
                                        extended-range redo (ctc_ext.h), the hand-over to the exact kernel
  scaled   144 (+ 48 wide alphabets)    the exact kernel's scaled form: blank-valued targets, targets beyond 447 labels,
                                        -inf log-probs, infeasible alignments
  exact    120                          E2E_ALGO_EXACT, f32 and f64, log domain
  fastauto 120                          unit-noise / trained-model emissions: the fast path proper (AUTO), ragged, any blank
  align    60                           e2e_ctc_align, bit-exact
The beam search's slice of the same kind lives in tests/test_gpu_beam.py::test_fuzz_slice_equals_the_oracle_or_is_a_proven_tie
(220 cases, run through both kernels).  The counts are printed in pytest's summary (tests/conftest.py).
"""
import ctypes
import threading

import numpy as np
import pytest
import torch

import oracle_lib as O
import gpu_util as U
from end2end_amd import _lib
from test_gpu_align import c_abi_align

pytestmark = pytest.mark.gpu

COUNTS = {}          # family -> [cases, utterances]; reported by conftest.pytest_terminal_summary


def _count(family, utterances):
    c = COUNTS.setdefault(family, [0, 0])
    c[0] += 1
    c[1] += utterances


def _feasible(tg, tl, xl):
    """shorten targets until every utterance has an alignment (the infeasible ones belong to the scaled / exact families)"""
    for b in range(len(tl)):
        need = lambda n: n + int((tg[b, 1:n] == tg[b, :n - 1]).sum())
        while need(int(tl[b])) > xl[b]:
            tl[b] = max(1, tl[b] // 2)


# ---------------------------------------------------------------------------------------------------------------------
# ext: tools/diag/fuzz_ext_vs_oracle.py
# ---------------------------------------------------------------------------------------------------------------------
def _ext_case(rng):
    B = int(rng.integers(1, 7))
    V = int(rng.choice([3, 5, 29, 29, 48, 80, 96, 97, 150, 224, 300, 448]))
    T = int(rng.choice([2, 9, 17, 64, 200, 500, 1000, 1000, 2100]))
    Smax = int(rng.integers(1, min(447, max(1, T // 2)) + 1))
    blank = int(rng.integers(0, V)) if rng.random() < 0.3 else 0
    mode = str(rng.choice(["sharp", "sharp", "peaky_noise", "neginf"]))
    labels = [v for v in range(V) if v != blank]
    tg = rng.choice(labels, size=(B, Smax))
    tl = rng.integers(1, Smax + 1, size=B)
    tl[0] = Smax
    xl = rng.integers(max(2, T // 2), T + 1, size=B)
    xl[0] = T
    _feasible(tg, tl, xl)
    logprobs = False
    if mode == "sharp":
        x = rng.standard_normal((B, T, V)) * float(rng.choice([3.0, 8.0, 12.0]))
    elif mode == "peaky_noise":
        x = rng.standard_normal((B, T, V))
        for b in range(B):
            src = (b + 1) % B if rng.random() < 0.5 else b        # half of the utterances emit ANOTHER one's transcript
            n2 = min(int(tl[src]), int(xl[b]))
            slots = np.sort(rng.choice(int(xl[b]), size=n2, replace=False))
            path = np.full(int(xl[b]), blank)
            path[slots] = tg[src, :n2]
            x[b, np.arange(int(xl[b])), path] += float(rng.choice([6.0, 10.0, 14.0]))
    else:
        x = rng.standard_normal((B, T, V)) * 8.0
        x[rng.random((B, T, V)) < 0.05] = -np.inf
        x[:, :, blank] = np.where(np.isinf(x[:, :, blank]), 0.0, x[:, :, blank])     # (the blank stays possible: a finite loss)
        logprobs = True
    return dict(x=x, tg=tg, xl=xl, tl=tl, blank=blank, logprobs=logprobs, mode=mode)


def _run_ext(c, io16):
    """-> None (equal), "marginal" (single elements within 1e-5: the f32 subtraction in the softmax at |x - max| up to 80,
    DESIGN 4.2a) or the assertion's text"""
    g_rtol, g_atol = (1e-4, 2e-6) if io16 is None else ((2.0 ** -7, 2.0 ** -8) if io16 == torch.bfloat16 else (2.0 ** -10, 2.0 ** -11))
    xt = torch.from_numpy(c["x"])
    tg, xl, tl, blank = c["tg"], c["xl"], c["tl"], c["blank"]
    if c["logprobs"]:
        arg = torch.log_softmax(xt, -1).float()
        l_o, g_o = O.ctc_loss(arg.double().numpy(), tg, xl, tl, blank)       # (the oracle on what the call is given: ROUNDED log-probs)
    else:
        arg = xt.float() if io16 is None else xt.to(io16)
        l_o, g_o = O.ctc_loss(torch.log_softmax(arg.double(), -1).numpy(), tg, xl, tl, blank)
        for b in range(len(xl)):
            g_o[b, xl[b]:] = 0.0
    la, ga = U.c_abi_loss(arg, tg, xl, tl, blank, c["logprobs"], _lib.ALGO_AUTO)
    try:
        U.assert_same(la, l_o, 1e-4, 2e-5, "losses")
        U.assert_same(ga, g_o, g_rtol, g_atol, "grads")
        return None
    except AssertionError as e:
        try:
            U.assert_same(la, l_o, 1e-4, 2e-5, "losses")
            U.assert_same(ga, g_o, g_rtol, max(1e-5, 2 * g_atol), "grads")
            return "marginal"
        except AssertionError:
            return str(e).strip().splitlines()[:6]


@pytest.mark.parametrize("seed,n,dtype", [(0, 150, None), (1, 150, None), (2, 70, torch.bfloat16), (3, 70, torch.float16)],
                         ids=["f32_seed0", "f32_seed1", "bf16_seed2", "f16_seed3"])
def test_ext_family_against_the_oracle(seed, n, dtype):
    rng = np.random.default_rng(seed)
    bad, marginal, done = [], [], 0
    for case in range(n):
        c = _ext_case(rng)
        if c["logprobs"] and dtype is not None:
            continue                                             # (16-bit log-probabilities cannot hold these: logits only)
        r = _run_ext(c, dtype)
        done += 1
        _count("ext", len(c["xl"]))
        if r == "marginal":
            marginal.append(case)
        elif r is not None:
            bad.append((case, c["mode"], c["x"].shape, int(c["tg"].shape[1]), c["blank"], r))
    assert not bad, "mismatching cases of seed %d: %s" % (seed, bad)
    assert len(marginal) <= max(1, done // 50), "marginal cases (one element within 1e-5) of seed %d: %s" % (seed, marginal)


# ---------------------------------------------------------------------------------------------------------------------
# scaled / exact: tools/diag/fuzz_scaled_vs_oracle.py, fuzz_exact_vs_oracle.py
# ---------------------------------------------------------------------------------------------------------------------
def _small_case(rng, case, family, wide):
    long_targets = family == "scaled" and case % 12 == 11
    if family == "scaled":
        B = int(rng.integers(1, 5))
        if long_targets:
            T = int(rng.integers(450, 700)); V = int(rng.choice([3, 29, 64])); Smax = int(rng.integers(448, min(T + 2, 600) + 1))
        else:
            T = int(rng.integers(1, 200)); V = int(rng.choice([2, 3, 5, 29, 64, 96, 200, 1500] if wide else [2, 3, 5, 29, 64, 96]))
            Smax = int(rng.integers(1, min(120, T + 3) + 1))
        f64 = False
        scale = float(rng.choice([0.3, 1.0, 4.0, 8.0, 12.0]))
    else:
        B = int(rng.integers(1, 6)); T = int(rng.integers(1, 90)); V = int(rng.choice([2, 3, 5, 29, 64, 97, 300]))
        Smax = int(rng.integers(0, min(60, T + 3) + 1))
        f64 = bool(rng.integers(0, 2))
        scale = float(rng.choice([0.3, 1.0, 4.0]))
    fused = bool(rng.integers(0, 2))
    blank = int(rng.choice([0, V - 1, rng.integers(0, V)]))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn(B, T, V, generator=g, dtype=torch.float64) * scale
    lp = torch.log_softmax(x, -1)
    if not fused and rng.integers(0, 3) == 0:
        lp[:, ::4, int(rng.integers(0, V))] = float("-inf")
    labs = [v for v in range(V) if v != blank]
    tg = torch.tensor(rng.choice(labs, size=(B, max(Smax, 1))), dtype=torch.long)
    xl = torch.tensor(rng.integers(1, T + 1, size=B)); xl[0] = T
    tl = torch.tensor(rng.integers(0, Smax + 1, size=B)); tl[0] = Smax
    if family == "scaled" and not long_targets:
        for b in range(B):
            if tl[b] > 0:
                tg[b, int(rng.integers(0, int(tl[b])))] = blank          # a target equal to the blank id: handed over by the fast path
    inp = (x if fused else lp).to(torch.float64 if f64 else torch.float32)
    return dict(inp=inp, fused=fused, tg=tg, xl=xl, tl=tl, blank=blank, f64=f64, shape=(B, T, V, Smax))


def _run_small(c, algo):
    ref_lp = torch.log_softmax(c["inp"].double(), -1) if c["fused"] else c["inp"].double()
    l_o, g_o = O.ctc_loss(ref_lp.numpy(), c["tg"].numpy(), c["xl"].numpy(), c["tl"].numpy(), c["blank"])
    if c["fused"]:
        for b in range(len(l_o)):
            if np.isfinite(l_o[b]):
                g_o[b, int(c["xl"][b]):] = 0.0          # (an infeasible utterance stays NaN everywhere, quirk Q2)
    lg, gg = U.c_abi_loss(c["inp"], c["tg"], c["xl"], c["tl"], c["blank"], not c["fused"], algo)
    rt, at = (1e-9, 1e-12) if c["f64"] else (1e-4, 2e-6)
    try:
        U.assert_same(lg, l_o, rt, at, "losses")
        U.assert_same(gg, g_o, rt, at, "grads")
        return None
    except AssertionError as e:
        return str(e).strip().splitlines()[:4]


@pytest.mark.parametrize("family,seed,n,wide", [("scaled", 0, 144, False), ("scaled", 5, 48, True), ("exact", 0, 120, False)],
                         ids=["scaled_seed0", "scaled_wide_seed5", "exact_seed0"])
def test_scaled_and_exact_families_against_the_oracle(family, seed, n, wide):
    rng = np.random.default_rng(seed)
    bad = []
    for case in range(n):
        c = _small_case(rng, case, family, wide)
        r = _run_small(c, _lib.ALGO_AUTO if family == "scaled" else _lib.ALGO_EXACT)
        _count(family, c["shape"][0])
        if r is not None:
            bad.append((case, c["shape"], c["fused"], c["blank"], c["f64"], r))
    assert not bad, "mismatching cases: %s" % bad


# ---------------------------------------------------------------------------------------------------------------------
# fastauto: what the fast path keeps (unit noise, a trained model's emissions), every lattice width, 32-bit and 16-bit I/O
# ---------------------------------------------------------------------------------------------------------------------
def _fast_case(rng):
    B = int(rng.integers(1, 9))
    V = int(rng.choice([2, 5, 29, 29, 32, 33, 64, 96, 120, 224, 448, 1000, 1001, 8000]))     # (1001: the wide path's two-pass rows, 16-bit too)
    T = int(rng.choice([1, 3, 15, 16, 17, 33, 100, 256, 511, 1000]))
    Smax = int(rng.integers(1, min(447, max(1, (T + 1) // 2)) + 1))
    if V >= 1000:
        Smax = min(Smax, 120)
    blank = int(rng.integers(0, V)) if rng.random() < 0.3 else 0
    labels = [v for v in range(V) if v != blank]
    tg = rng.choice(labels, size=(B, Smax))
    if rng.random() < 0.3 and Smax > 1:
        tg[:, 1::2] = tg[:, 0::2][:, :tg[:, 1::2].shape[1]]                     # adjacent repeats
    tl = rng.integers(1, Smax + 1, size=B); tl[0] = Smax
    xl = rng.integers(max(1, T // 2), T + 1, size=B); xl[0] = T
    _feasible(tg, tl, xl)
    x = rng.standard_normal((B, T, V)) * float(rng.choice([0.1, 1.0, 1.0, 2.0]))
    if rng.random() < 0.4:
        for b in range(B):                                                       # a trained model: the own transcript boosted
            n = min(int(tl[b]), int(xl[b]))
            slots = np.sort(rng.choice(int(xl[b]), size=n, replace=False))
            x[b, slots, tg[b, :n]] += float(rng.choice([4.0, 8.0]))
    logprobs = bool(rng.random() < 0.3)
    time_major = bool(rng.random() < 0.3)
    return dict(x=x, tg=tg, xl=xl, tl=tl, blank=blank, logprobs=logprobs, time_major=time_major)


@pytest.mark.parametrize("seed,n", [(10, 60), (11, 60)], ids=["seed10", "seed11"])
def test_fast_path_family_against_the_oracle(seed, n):
    rng = np.random.default_rng(seed)
    bad = []
    for case in range(n):
        c = _fast_case(rng)
        xt = torch.from_numpy(c["x"])
        io16 = [None, None, torch.bfloat16, torch.float16][case % 4]
        if c["logprobs"]:
            io16 = None
        arg = xt.float() if io16 is None else xt.to(io16)
        lp = torch.log_softmax(arg.double(), -1)
        if c["logprobs"]:
            arg = lp.float(); lp = arg.double()
        l_o, g_o = O.ctc_loss(lp.numpy(), c["tg"], c["xl"], c["tl"], c["blank"])
        if not c["logprobs"]:
            for b in range(len(c["xl"])):
                g_o[b, c["xl"][b]:] = 0.0
        if c["time_major"]:
            arg = arg.permute(1, 0, 2).contiguous().permute(1, 0, 2)            # a time-major view, no copy
        la, ga = U.c_abi_loss(arg, c["tg"], c["xl"], c["tl"], c["blank"], c["logprobs"], _lib.ALGO_AUTO)
        g_rtol, g_atol = (1e-4, 2e-6) if io16 is None else ((2.0 ** -7, 2.0 ** -8) if io16 == torch.bfloat16 else (2.0 ** -10, 2.0 ** -11))
        _count("fastauto", len(c["xl"]))
        try:
            U.assert_same(la, l_o, 1e-4, 2e-5, "losses")
            U.assert_same(ga, g_o, g_rtol, g_atol, "grads")
        except AssertionError as e:
            bad.append((case, c["x"].shape, int(c["tg"].shape[1]), c["blank"], str(io16), c["logprobs"], str(e).strip().splitlines()[:4]))
    assert not bad, "mismatching cases: %s" % bad


# ---------------------------------------------------------------------------------------------------------------------
# align: tools/diag/fuzz_align.py (bit-exact)
# ---------------------------------------------------------------------------------------------------------------------
def test_align_family_is_bit_exact_against_the_oracle():
    rng = np.random.default_rng(0)
    bad = []
    for case in range(60):
        B = int(rng.integers(1, 6)); T = int(rng.integers(1, 300)); V = int(rng.integers(2, 40))
        S = int(rng.integers(0, min(T, 120) + 1))
        is_ctc = bool(rng.integers(0, 2))
        blank = int(rng.choice([0, V - 1, rng.integers(0, V)])) if is_ctc else 0
        g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        x = torch.randn(B, T, V, generator=g, dtype=torch.float64) * float(rng.choice([0.0, 0.3, 1.0, 4.0]))
        style = int(rng.integers(0, 3))
        if style == 1:
            x = x.round()
        lp = torch.log_softmax(x, -1)
        if style == 2 and V > 2:
            lp[:, ::3, int(rng.integers(0, V))] = float("-inf")
        pool = [v for v in range(V) if v != blank] if is_ctc else list(range(V))
        tg = torch.tensor(rng.choice(pool, size=(B, max(S, 1))))
        if rng.integers(0, 2) and S > 1:
            tg[:, 1::2] = tg[:, 0::2][:, : tg[:, 1::2].shape[1]]
        tl = rng.integers(0 if is_ctc else 1, S + 1, size=B) if S > 0 else np.zeros(B, dtype=np.int64)
        if not is_ctc:
            tl = np.maximum(tl, 1)
        xl = rng.integers(1, T + 1, size=B); xl[0] = T
        if not is_ctc:
            xl = np.maximum(xl, np.minimum(tl, T))
        if rng.integers(0, 2):
            lp = lp.float()
        got = c_abi_align(lp, tg, xl, tl, blank, is_ctc)
        want = O.ctc_align(lp.double().numpy(), tg.numpy(), xl, tl, blank, is_ctc)
        _count("align", B)
        if not np.array_equal(got, want):
            bad.append((case, dict(B=B, T=T, V=V, S=S, is_ctc=is_ctc, blank=blank, style=style)))
    assert not bad, bad


# ---------------------------------------------------------------------------------------------------------------------
# Round 5's regressions as named cases (the other two: test_gpu_regimes.py::test_probabilities_at_the_end_of_f32_...,
# test_gpu_loss.py::test_long_transcripts_take_the_fast_path[B2_T1000_V96_S425 / B2_T900_V80_S300])
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("V,S,T", [(29, 200, 1000), (29, 100, 2100), (48, 223, 1000)], ids=lambda v: str(v))
def test_no_redo_reports_success_over_nan_rows(V, S, T):
    """Sharp unrelated emissions at scale 12: row sums of the f64 segment redo sit near the end of f64, where the reciprocal of
    its Newton step overflowed -- NaN rows were written and the redo reported success (rounds 3..5).  Whatever route an
    utterance takes, a finite oracle loss means a finite gradient slab, equal to the oracle's."""
    rng = np.random.default_rng(V + S)
    B = 4
    x = (rng.standard_normal((B, T, V)) * 12.0).astype(np.float32)
    tg = rng.integers(1, V, size=(B, S)); tl = rng.integers(S // 2, S + 1, size=B); tl[0] = S
    xl = np.array([T, T - 7, T // 2 + 3, T])
    lp = torch.log_softmax(torch.from_numpy(x).double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg, xl, tl, 0)
    for b in range(B):
        g_o[b, xl[b]:] = 0.0
    la, ga = U.c_abi_loss(torch.from_numpy(x), tg, xl, tl, 0, False, _lib.ALGO_AUTO)
    assert np.isfinite(l_o).all() and np.isfinite(la).all() and np.isfinite(ga).all()
    U.assert_same(la, l_o, 1e-4, 2e-5, "losses")
    U.assert_same(ga, g_o, 1e-4, 1e-5, "grads")          # (scale 12: single elements up to 7e-6)


# ---------------------------------------------------------------------------------------------------------------------
# The flagged launch's bounded waits under contention
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("lds", [0, 120 * 1024], ids=["issue_slots_only", "lds_taken"])
def test_flagged_batch_is_right_while_another_stream_holds_the_cus(lds):
    """The flagged launch is a multi-round protocol between the workgroups of ONE launch with bounded waits; the regime tests
    assert that no wait runs out on an idle GPU.  Here a second stream keeps the chip busy with long-running workgroups
    (`e2e_debug_occupy` from the same library: 224 workgroups of 1024 threads resident for 3 ms each, launched back to back;
    with 120 KB of LDS each no workgroup of the flagged launch fits beside one, so part of its grid is NOT resident while the
    rest waits) while a batch with mislabelled utterances, range-flagged utterances and a blank-valued target runs.  The
    results must equal the oracle's WHATEVER the counters say -- a wait that runs out has to end in the slow route, not in a
    wrong or unsettled row."""
    L = _lib.load()
    if not hasattr(L, "e2e_debug_occupy"):
        pytest.skip("library without e2e_debug_occupy")
    L.e2e_debug_occupy.restype = ctypes.c_int
    L.e2e_debug_occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
    d = U.dev()
    from test_gpu_regimes import aligned
    rng = np.random.default_rng(77)
    x, tg, xl, tl = aligned(rng, 12, 1000, 29, 200, 10.0, short=[(5, 933)])
    tg[1], tl[1] = tg[2].copy(), tl[2]                          # mislabelled: extended-range chains
    tg[5], tl[5] = tg[6].copy(), tl[6]
    x[7] = rng.standard_normal(x[7].shape) * 3.0                # sharp unrelated at scale 3: range flags, f64 segment redo
    x[8] = rng.standard_normal(x[8].shape) * 3.0
    tg[9, 3] = 0                                                # a target equal to the blank: full recomputation (step 2)
    lp = torch.log_softmax(torch.from_numpy(x).double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg, xl, tl, 0)
    for b in range(len(xl)):
        g_o[b, xl[b]:] = 0.0
    side = torch.cuda.Stream(device=d)
    stop = threading.Event()

    def hog():
        torch.cuda.set_device(d)
        while not stop.is_set():
            for _ in range(4):
                _lib.check(L.e2e_debug_occupy(224, 1024, lds, 3_000_000, ctypes.c_void_p(side.cuda_stream)))
            side.synchronize()

    th = threading.Thread(target=hog)
    th.start()
    try:
        outcomes = []
        for rep in range(6):
            keep = {}
            la, ga = U.c_abi_loss(torch.from_numpy(x), tg, xl, tl, 0, False, _lib.ALGO_AUTO, keep=keep)
            U.assert_same(la, l_o, 1e-4, 2e-5, "losses (rep %d)" % rep)
            U.assert_same(ga, g_o, 1e-4, 2e-6, "grads (rep %d)" % rep)
            to, fr = ctypes.c_int(-1), ctypes.c_int(-1)
            L.e2e_debug_flagged_counters.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
            L.e2e_debug_flagged_counters(keep["workspace"].data_ptr(), 12, 1000, 29, 200, ctypes.byref(to), ctypes.byref(fr))
            outcomes.append((to.value, fr.value))
    finally:
        stop.set()
        th.join()
    torch.cuda.synchronize()
    print("flagged launch under contention: (waits that ran out, failed redos) per repetition:", outcomes)


@pytest.mark.parametrize("shape", [(300, 70, 66, 24), (520, 40, 90, 12)], ids=lambda s: "B%d_T%d_V%d_S%d" % s)
def test_batches_beyond_the_cu_count_take_the_shallow_probability_ring(shape):
    """More utterances than the chip has CUs on an alphabet whose f64 probability ring of eight blocks keeps a second workgroup off
    the CU (61..96 columns): the single-wave chain kernel then runs with a ring of four (round 6; BASELINE configs[4]'s compact
    lattice is the case that matters).  Ragged lengths, repeats, against the oracle."""
    B, T, V, S = shape
    rng = np.random.default_rng(B)
    x = rng.standard_normal((B, T, V)).astype(np.float32)
    tg = rng.integers(1, V, size=(B, S))
    tg[:, 1::3] = tg[:, 0::3][:, :tg[:, 1::3].shape[1]]
    tl = rng.integers(1, S + 1, size=B); tl[0] = S
    xl = rng.integers(2 * S + 2, T + 1, size=B); xl[0] = T
    lp = torch.log_softmax(torch.from_numpy(x).double(), -1).numpy()
    l_o, g_o = O.ctc_loss(lp, tg, xl, tl, 0)
    for b in range(B):
        g_o[b, xl[b]:] = 0.0
    la, ga = U.c_abi_loss(torch.from_numpy(x), tg, xl, tl, 0, False, _lib.ALGO_AUTO)
    U.assert_same(la, l_o, 1e-4, 2e-5, "losses")
    U.assert_same(ga, g_o, 1e-4, 2e-6, "grads")
    lf, _ = U.c_abi_loss(torch.from_numpy(x), tg, xl, tl, 0, False, _lib.ALGO_FAST)
    assert not np.isnan(lf).any(), "the fast path gave up on %s" % np.nonzero(np.isnan(lf))[0].tolist()
