"""-m gpu: small randomised sweeps (fixed seeds) of the fast path against the exact kernel, which the other tests pin to
the oracle.  The long versions live in tools/diag/fuzz_*.py; a sweep like this found the dense-target bug of round 1."""
import numpy as np
import pytest
import torch

import gpu_util as U
from end2end_amd import _lib

pytestmark = pytest.mark.gpu


def _sweep(seed, n_cases, mode, chains=None, grad_atol=2e-6):
    """chains: e2e_ctc_loss_opts.chains for the fast-path calls (None: the plain entry point)."""
    rng = np.random.default_rng(seed)
    compared = 0
    for _ in range(n_cases):
        B = int(rng.integers(1, 7)); T = int(rng.integers(1, 500)); V = int(rng.integers(2, 97))
        if mode == "wide":
            V = int(rng.choice([97, 200, 1500])); T = int(rng.integers(1, 120))
        Smax = int(rng.integers(0, min(255, T) + 1))
        if mode == "dense" and T > 4:
            Smax = int(min(255, max(1, T * rng.uniform(0.45, 0.98))))
        if mode == "edges":
            T = int(rng.integers(1, 48)) if rng.integers(0, 2) else int(rng.integers(130, 400))
            Smax = int(min(T, rng.choice([0, 1, 62, 63, 64, 65, 126, 127, 128, 129, 254, 255])))
        if mode == "wide":
            Smax = min(Smax, 90)
        sharp = float(rng.choice([3.0, 5.0, 8.0])) if mode == "sharp" else float(rng.choice([0.1, 1.0, 3.0]))
        fused = bool(rng.integers(0, 2)); blank = int(rng.choice([0, V - 1, rng.integers(0, V)]))
        g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        x = torch.randn(B, T, V, generator=g) * sharp
        if not fused:
            x = torch.log_softmax(x.double(), -1).float()
        labs = [v for v in range(V) if v != blank]
        tg = torch.tensor(rng.choice(labs, size=(B, max(Smax, 1))), dtype=torch.long)
        xl = torch.tensor(rng.integers(1, T + 1, size=B)); xl[0] = T
        tl = torch.tensor(rng.integers(Smax // 2 if mode == "dense" else 0, Smax + 1, size=B)); tl[0] = Smax
        le, ge = U.c_abi_loss(x, tg, xl, tl, blank, not fused, _lib.ALGO_EXACT)
        lf, gf = U.c_abi_loss(x, tg, xl, tl, blank, not fused, _lib.ALGO_FAST, chains=chains)
        if mode == "sharp":
            # the default path -- flagged utterances redone from the chains' checkpoints in f64, or by the exact kernel
            la, ga = U.c_abi_loss(x, tg, xl, tl, blank, not fused, _lib.ALGO_AUTO, chains=chains)
            for b in range(B):
                if np.isfinite(le[b]):
                    assert abs(float(la[b]) - float(le[b])) <= 1e-4 * max(1.0, abs(float(le[b]))), (mode, B, T, V, Smax, b)
                    np.testing.assert_allclose(ga[b], ge[b], rtol=1e-4, atol=grad_atol)
        for b in range(B):
            if np.isnan(lf[b]):
                continue                                     # the fast path gave up: AUTO would take the exact result
            compared += 1
            what = "mode %s B=%d T=%d V=%d S=%d fused=%d blank=%d utt %d (xl=%d tl=%d)" % (
                mode, B, T, V, Smax, fused, blank, b, int(xl[b]), int(tl[b]))
            assert abs(float(lf[b]) - float(le[b])) <= 1e-4 * max(1.0, abs(float(le[b]))), what
            np.testing.assert_allclose(gf[b], ge[b], rtol=1e-4, atol=grad_atol, err_msg=what)
    return compared


@pytest.mark.parametrize("mode,seed,n", [("general", 101, 40), ("dense", 102, 40), ("edges", 103, 60), ("wide", 104, 25)])
def test_fast_path_agrees_with_exact_kernel(mode, seed, n):
    assert _sweep(seed, n, mode) > n // 2


def test_sharp_unrelated_emissions_are_taken_or_handed_over_but_never_wrong():
    """Logits at scale 3 .. 8 against random targets: the regime where rows of the lattice outgrow f32 and the segment
    kernel's partition-sum self-check decides what the fast path may keep.  Whatever it keeps must be right, and what the
    caller gets by default (ALGO_AUTO) must be right for every utterance."""
    assert _sweep(105, 40, "sharp") > 0


# e2e_ctc_loss_opts.chains = E2E_CHAINS_F32: the chains of long-target utterances in packed f32.  The tolerance of the
# gradient elements is the one include/e2e_ctc.h states for the option (2e-5 absolute next to 1e-4 relative); losses as ever.
F32_CHAINS_GRAD_ATOL = 2e-5


@pytest.mark.parametrize("mode,seed,n", [("general", 201, 40), ("dense", 202, 40), ("edges", 203, 60), ("sharp", 205, 30)])
def test_f32_chains_option_agrees_with_exact_kernel(mode, seed, n):
    assert _sweep(seed, n, mode, chains=_lib.CHAINS_F32, grad_atol=F32_CHAINS_GRAD_ATOL) > (0 if mode == "sharp" else n // 2)


def test_f32_chains_option_changes_nothing_where_it_does_not_apply():
    """Short targets (rows of <= 256 cells) and wide alphabets keep the f64 chains: bit-identical results."""
    g = torch.Generator().manual_seed(9)
    for (B, T, V, S) in ((4, 200, 29, 60), (3, 150, 29, 127), (2, 60, 300, 20)):
        x = torch.randn(B, T, V, generator=g)
        tg = torch.randint(1, V, (B, S), generator=g)
        xl = torch.full((B,), T); tl = torch.randint(S // 2, S + 1, (B,), generator=g)
        l0, g0 = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_AUTO)
        l1, g1 = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_AUTO, chains=_lib.CHAINS_F32)
        assert np.array_equal(l0, l1) and np.array_equal(g0, g1)


def test_bad_chains_value_is_refused():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 8, 5, generator=g)
    with pytest.raises(_lib.E2EError, match="chains"):
        U.c_abi_loss(x, torch.tensor([[1, 2]]), torch.tensor([8]), torch.tensor([2]), 0, False, _lib.ALGO_AUTO, chains=7)


def _variant_in_child(env_name, grad_atol):
    """A chain-kernel variant that is selected by an environment variable read once per process: run in a child."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import numpy as np, torch
import gpu_util as U
from end2end_amd import _lib
g = torch.Generator().manual_seed(77)
for (B, T, V, S) in ((6, 300, 29, 100), (4, 1000, 29, 200), (5, 137, 20, 60), (3, 64, 9, 31), (3, 520, 40, 128), (3, 700, 64, 255), (2, 33, 4, 0)):
    x = torch.randn(B, T, V, generator=g) * 1.5
    tg = torch.randint(1, V, (B, max(S, 1)), generator=g)
    xl = torch.randint(max(2 * S + 1, T // 2), T + 1, (B,), generator=g); xl[0] = T
    tl = torch.randint(S // 2, S + 1, (B,), generator=g); tl[0] = S
    le, ge = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_EXACT)
    lf, gf = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_FAST)
    assert np.isfinite(lf).all(), lf
    U.assert_same(lf, le, 2e-6, 1e-6, "losses")
    U.assert_same(gf, ge, 1e-4, GRAD_ATOL, "grads")
print("ok")
'''.replace("GRAD_ATOL", repr(grad_atol))
    env = dict(os.environ, **{env_name: "1"})
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_f32_chains_forced_for_every_row_width_match_exact():
    """E2E_F1_F32=1 sends every shape through the f32 chain kernel (1, 2 and 4 label pairs per segment-kernel lane)."""
    _variant_in_child("E2E_F1_F32", F32_CHAINS_GRAD_ATOL)


def test_single_wave_chains_forced_for_every_row_width_match_exact():
    """E2E_F1_SINGLE=1: the single-wave f64 chains also where the halo chains are the default (targets of 128..223 labels)."""
    _variant_in_child("E2E_F1_SINGLE", 2e-6)


def test_f64_halo_chains_forced_wherever_they_fit_match_exact():
    """E2E_F1_HALO=1: the f64 halo chains for every row width up to 223 labels (default only for the widest rows)."""
    _variant_in_child("E2E_F1_HALO", 2e-6)


def test_flagged_utterances_are_settled_by_the_segment_redo_not_by_the_exact_kernel():
    """The headline shape with emissions that contradict the targets (logits x3): a third of the utterances leave the
    f32 segment kernel's range.  They must be redone from the chains' checkpoints in f64 (~1 ms for the batch), not by
    the exact kernel (~7 ms): a regression here is invisible in the results and shows only as a step-time cliff."""
    import ctypes
    L = _lib.load()
    g = torch.Generator().manual_seed(3)
    B, T, V, S = 64, 1000, 29, 200
    x = (torch.randn(B, T, V, generator=g) * 3.0)
    tg = torch.randint(1, V, (B, S), generator=g)
    xl = torch.full((B,), T); tl = torch.randint(S // 2, S + 1, (B,), generator=g)
    kept = {}
    lf, _ = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_FAST)
    la, _ = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_AUTO, keep=kept)
    flagged = int(np.isnan(lf).sum())
    assert flagged >= 4 and np.isfinite(la).all()
    n = ctypes.c_int(-1)
    L.e2e_debug_fast_redo_failures.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    assert L.e2e_debug_fast_redo_failures(kept["workspace"].data_ptr(), B, T, V, S, ctypes.byref(n)) == 0
    assert 0 <= n.value <= flagged // 4, (n.value, flagged)


@pytest.mark.parametrize("seed,n", [(301, 60), (302, 60)])
def test_scaled_exact_form_agrees_with_the_log_domain_kernel(seed, n):
    """Under AUTO with f32 the exact kernel walks in the probability domain (ctc_loss_exact.hip, SCALED).  Every utterance here
    reaches it -- a target equal to the blank id in the batch's every utterance that has targets (the fast path hands those
    over), or targets beyond the fast kernels' 447 labels -- and must agree with E2E_ALGO_EXACT, the reference's log-domain
    arithmetic (pinned to the oracle elsewhere), to f32 rounding: T from 1, empty targets, ragged lengths, repeats,
    infeasible alignments (the log-domain walk's inf / NaN pattern), fused logits and log-probs, any blank id."""
    rng = np.random.default_rng(seed)
    for case in range(n):
        long_targets = case % 6 == 5
        B = int(rng.integers(1, 5))
        if long_targets:
            T = int(rng.integers(460, 1200)); V = int(rng.integers(3, 60)); Smax = int(rng.integers(448, min(T, 640) + 1))
        else:
            T = int(rng.integers(1, 300)); V = int(rng.integers(2, 97)); Smax = int(rng.integers(1, min(255, T) + 1))
        sharp = float(rng.choice([0.1, 1.0, 3.0, 8.0]))
        fused = bool(rng.integers(0, 2)); blank = int(rng.choice([0, V - 1, rng.integers(0, V)]))
        g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        x = torch.randn(B, T, V, generator=g) * sharp
        if not fused:
            x = torch.log_softmax(x.double(), -1).float()
        labs = [v for v in range(V) if v != blank]
        few = labs[: max(1, len(labs) // 8)] if rng.integers(0, 2) else labs        # (few labels: many adjacent repeats)
        tg = torch.tensor(rng.choice(few, size=(B, Smax)), dtype=torch.long)
        xl = torch.tensor(rng.integers(1, T + 1, size=B)); xl[0] = T
        tl = torch.tensor(rng.integers(0, Smax + 1, size=B)); tl[0] = Smax
        if not long_targets:
            for b in range(B):
                if tl[b] > 0:
                    tg[b, int(rng.integers(0, int(tl[b])))] = blank             # handed over by the fast path
        le, ge = U.c_abi_loss(x, tg, xl, tl, blank, not fused, _lib.ALGO_EXACT)
        la, ga = U.c_abi_loss(x, tg, xl, tl, blank, not fused, _lib.ALGO_AUTO)
        what = "case %d: B=%d T=%d V=%d S=%d fused=%d blank=%d sharp=%g xl=%s tl=%s" % (
            case, B, T, V, Smax, fused, blank, sharp, xl.tolist(), tl.tolist())
        U.assert_same(la, le, 2e-6, 0, "losses, " + what)
        U.assert_same(ga, ge, 1e-5, 2e-7, "grads, " + what)
