"""Randomised parity sweep of the extended-range redo (end2end_amd/csrc/ctc_ext.h) against the oracle: E2E_ALGO_AUTO on emissions
that drive utterances off the f32 lattice -- sharp unrelated logits of scale 3 .. 12, peaky emissions with mislabelled
utterances, -inf log-probabilities --, T from 2 to 2100, targets of 1 .. 447 labels (one and two pairs per lane of the chains),
alphabets of 3 .. 448 columns (both layouts of the probability table), ragged lengths, any blank id, mixed batches.
Prints how many utterances the fast path kept, how many the extended-range redo settled and how many the exact kernel had to.
  python tools/diag/fuzz_ext_vs_oracle.py [cases] [seed]          FUZZ_DTYPE=bf16 / f16: 16-bit logits (gradient at the type's resolution)"""
import ctypes, sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import gpu_util as U, oracle_lib as O
from end2end_amd import _lib
L = _lib.load()
L.e2e_debug_fast_redo_failures.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0; tot = 0; off_fast = 0; exact = 0; marginal = 0
io16 = {"bf16": torch.bfloat16, "f16": torch.float16}.get(os.environ.get("FUZZ_DTYPE", ""))
g_rtol, g_atol = (1e-4, 2e-6) if io16 is None else ((2.0 ** -7, 2.0 ** -8) if io16 == torch.bfloat16 else (2.0 ** -10, 2.0 ** -11))
for case in range(n_cases):
    B = int(rng.integers(1, 7))
    V = int(rng.choice([3, 5, 29, 29, 48, 80, 96, 97, 150, 224, 300, 448]))
    T = int(rng.choice([2, 9, 17, 64, 200, 500, 1000, 1000, 2100]))
    Smax = int(rng.integers(1, min(447, max(1, T // 2)) + 1))
    blank = int(rng.integers(0, V)) if rng.random() < 0.3 else 0
    mode = rng.choice(["sharp", "sharp", "peaky_noise", "neginf"])
    labels = [v for v in range(V) if v != blank]
    tg = rng.choice(labels, size=(B, Smax)); tl = rng.integers(1, Smax + 1, size=B); tl[0] = Smax
    xl = rng.integers(max(2, T // 2), T + 1, size=B); xl[0] = T
    for b in range(B):     # keep every utterance feasible (the infeasible ones are the exact kernel's business, tested elsewhere)
        need = int(tl[b]) + int((tg[b, 1:tl[b]] == tg[b, :tl[b] - 1]).sum())
        while need > xl[b]:
            tl[b] = max(1, tl[b] // 2); need = int(tl[b]) + int((tg[b, 1:tl[b]] == tg[b, :tl[b] - 1]).sum())
    logprobs = False
    if mode == "sharp":
        x = rng.standard_normal((B, T, V)) * float(rng.choice([3.0, 8.0, 12.0]))
    elif mode == "peaky_noise":
        x = rng.standard_normal((B, T, V))
        for b in range(B):
            n = int(tl[b]); src = (b + 1) % B if rng.random() < 0.5 else b        # half of the utterances emit ANOTHER one's transcript
            n2 = min(int(tl[src]), int(xl[b]))
            slots = np.sort(rng.choice(int(xl[b]), size=n2, replace=False)); path = np.full(int(xl[b]), blank); path[slots] = tg[src, :n2]
            x[b, np.arange(int(xl[b])), path] += float(rng.choice([6.0, 10.0, 14.0]))
    else:
        x = rng.standard_normal((B, T, V)) * 8.0
        x[rng.random((B, T, V)) < 0.05] = -np.inf
        x[:, :, blank] = np.where(np.isinf(x[:, :, blank]), 0.0, x[:, :, blank])     # (the blank stays possible: a finite loss)
        logprobs = True
    if os.environ.get("FUZZ_ONLY") and case != int(os.environ["FUZZ_ONLY"]): continue
    xt = torch.from_numpy(x)
    lp = torch.log_softmax(xt, -1)
    l_o, g_o = O.ctc_loss(lp.numpy(), tg, xl, tl, blank)
    if logprobs and io16 is not None: continue                  # (16-bit log-probabilities cannot hold these: logits only)
    if logprobs:
        arg = lp.float()
        l_o, g_o = O.ctc_loss(arg.double().numpy(), tg, xl, tl, blank)       # (the oracle on what the call is given: the ROUNDED log-probabilities)
    else:
        arg = xt.float() if io16 is None else xt.to(io16)        # (FUZZ_DTYPE=bf16 / f16: 16-bit logits in, 16-bit gradient out)
        lp32 = torch.log_softmax(arg.double(), -1); l_o, g_o = O.ctc_loss(lp32.numpy(), tg, xl, tl, blank)
        for b in range(B): g_o[b, xl[b]:] = 0.0
    lf, _ = U.c_abi_loss(arg, tg, xl, tl, blank, logprobs, _lib.ALGO_FAST) if _lib.load().e2e_ctc_loss_workspace_bytes(B, T, V, Smax, 0, _lib.ALGO_FAST) else (np.full(B, np.nan), None)
    keep = {}
    la, ga = U.c_abi_loss(arg, tg, xl, tl, blank, logprobs, _lib.ALGO_AUTO, keep=keep)
    cnt = ctypes.c_int(0)
    L.e2e_debug_fast_redo_failures(keep["workspace"].data_ptr(), B, T, V, Smax, ctypes.byref(cnt))
    tot += B; off_fast += int(np.isnan(lf).sum()); exact += cnt.value
    try:
        U.assert_same(la, l_o, 1e-4, 2e-5, "losses"); U.assert_same(ga, g_o, g_rtol, g_atol, "grads")
    except AssertionError as e:
        # (marginal: single elements off by < 1e-5 at emissions of scale 8 .. 12.  The softmax's x - max is an f32 subtraction: with
        #  |x - max| up to 80 its half ulp is 3.8e-6 in the exponent, i.e. 4e-6 relative in a probability, and two alignments that
        #  compete for a frame's posterior differ in dozens of such factors -- tools/diag/ext_case_row.py shows the mass moved between
        #  two columns of a row whose sum is 1 to 1e-7.  The reference reads f32 log-probabilities, which carry the same rounding;
        #  the oracle here takes the softmax in f64.  The wide-row table kernel carries the subtraction's error along since the end
        #  of round 5 (two-sum); the narrow chains' producers do not.)
        try:
            U.assert_same(la, l_o, 1e-4, 2e-5, "losses"); U.assert_same(ga, g_o, g_rtol, max(1e-5, 2 * g_atol), "grads"); marginal += 1
        except AssertionError:
            bad += 1
        if os.environ.get("FUZZ_ONLY"):
            fl = (ctypes.c_int * B)(); lz = (ctypes.c_double * (2 * B))()
            L.e2e_debug_fast_state.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
            L.e2e_debug_fast_state(keep["workspace"].data_ptr(), B, T, V, Smax, fl, lz)
            print(" flag words:", list(fl))
            for b in range(B):
                nn = np.argwhere(np.isnan(ga[b]) != np.isnan(g_o[b]))
                d = np.abs(np.nan_to_num(ga[b].astype(np.float64)) - np.nan_to_num(g_o[b]))
                if len(nn): print("   rows with NaN:", sorted(set(nn[:, 0].tolist())), "cols in first such row:", nn[nn[:, 0] == nn[0, 0], 1].tolist())
                print(" utt %d: T=%d S=%d loss %r oracle %r; NaN-pattern differences %d (first %s); max |dgrad| %.2e at %s" % (
                    b, xl[b], tl[b], la[b], l_o[b], len(nn), nn[:4].tolist(), d.max(), np.unravel_index(d.argmax(), d.shape)))
        print("case %d MISMATCH: mode %s B=%d T=%d V=%d Smax=%d blank=%d xl=%s tl=%s fast-flagged %s exact %d: %s" % (
            case, mode, B, T, V, Smax, blank, xl.tolist(), tl.tolist(), np.nonzero(np.isnan(lf))[0].tolist(), cnt.value, str(e).splitlines()[0:6]), flush=True)
print("%d cases, %d utterances: %d left the f32 lattice, %d of them were recomputed by the exact kernel; %d mismatching cases (+ %d marginal: one element within 1e-5)" % (n_cases, tot, off_fast, exact, bad, marginal))
