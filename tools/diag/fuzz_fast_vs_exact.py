"""Randomised parity sweep on the GPU: the fast path (E2E_ALGO_FAST, no fallback) against the exact f64-log-domain kernel
(E2E_ALGO_EXACT, itself pinned to the oracle by the tests) over random shapes, lengths, alphabets, blank positions and
emission sharpness.  Utterances the fast path flags (NaN) are counted, not compared.  Prints the worst deviations."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import gpu_util as U
from end2end_amd import _lib
if os.environ.get('E2E_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['E2E_LIB'])
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
mode = sys.argv[3] if len(sys.argv) > 3 else ''
dense = mode == 'dense'        # target lengths close to the input lengths
edges = mode == 'edges'        # short inputs, target lengths around the lane-packing boundaries
wide = mode == 'wide'          # alphabets beyond the lattice kernels' columns (compaction path)
mid = mode in ('mid', 'middense')   # alphabets of 97..448 columns: the lattice kernels' wide-row forms, no compaction
verbose = os.environ.get('FUZZ_VERBOSE') == '1'
check_auto = os.environ.get('FUZZ_AUTO', '1') == '1'
auto_bad = 0
import ctypes
zdev_good, zdev_bad = [], []       # instrumented library only: the segment self-check's deviation, by outcome
worst_l, worst_g, flagged, total = 0.0, 0.0, 0, 0
import collections
tally = collections.Counter(); feas = collections.Counter(); flagwords = collections.Counter()
for case in range(n_cases):
    B = int(rng.integers(1, 9)); T = int(rng.integers(1, 700)); V = int(rng.integers(2, 97))
    if wide: V = int(rng.choice([230, 500, 1000, 3001])); T = int(rng.integers(1, 300))
    if mid: V = int(rng.integers(97, 449)); T = int(rng.integers(1, 700))
    if edges and rng.integers(0, 2): T = int(rng.integers(1, 48))
    Smax = int(rng.integers(0, min(447 if (mid or wide) else 255, T) + 1))
    if mode == 'middense' and T > 4: Smax = int(min(447, max(1, T * rng.uniform(0.45, 0.98))))
    if edges and T >= 70: Smax = int(min(T, rng.choice([62, 63, 64, 65, 126, 127, 128, 129, 254, 255])))
    if dense and T > 4: Smax = int(min(255, max(1, T * rng.uniform(0.45, 0.98))))
    sharp = float(rng.choice([0.1, 1.0, 3.0] if dense else [0.1, 1.0, 3.0, 8.0, 20.0]))
    fused = bool(rng.integers(0, 2)); blank = int(rng.choice([0, V - 1, rng.integers(0, V)]))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn(B, T, V, generator=g) * sharp
    if not fused: x = torch.log_softmax(x.double(), -1).float()
    labs = [v for v in range(V) if v != blank]
    tg = torch.tensor(rng.choice(labs, size=(B, max(Smax, 1))), dtype=torch.long)
    xl = torch.tensor(rng.integers(1, T + 1, size=B)); xl[0] = T
    tl = torch.tensor(rng.integers(Smax // 2 if dense else 0, Smax + 1, size=B)); tl[0] = Smax
    le, ge = U.c_abi_loss(x, tg[:, :max(Smax, 1)], xl, tl, blank, not fused, _lib.ALGO_EXACT)
    has_zdev = bool(os.environ.get('E2E_LIB')) and hasattr(_lib.load(), 'e2e_debug_fast_zdev')
    if has_zdev: _lib.load().e2e_debug_fast_zdev(None, 1)
    kept = {}
    lf, gf = U.c_abi_loss(x, tg[:, :max(Smax, 1)], xl, tl, blank, not fused, _lib.ALGO_FAST, keep=kept)
    if V <= 96:
        fw = (ctypes.c_int * B)(); lzw = (ctypes.c_double * (2 * B))()
        _lib.load().e2e_debug_fast_state.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
        if _lib.load().e2e_debug_fast_state(kept["workspace"].data_ptr(), B, T, V, max(Smax, 1), fw, lzw) == 0:
            for b in range(B):
                if np.isfinite(le[b]): flagwords[(sharp, fw[b])] += 1
    zd = None
    if has_zdev and V <= 96 and B * ((T + 15) // 16) <= 16384:
        buf = np.zeros(16384, np.float32)
        _lib.load().e2e_debug_fast_zdev(buf.ctypes.data_as(ctypes.c_void_p), 0)
        zd = buf[:B * ((T + 15) // 16)].reshape(B, -1)
    if check_auto:
        # what the caller gets by default: flagged utterances redone (f64 segments, or the exact kernel) -- every one must agree
        la, ga = U.c_abi_loss(x, tg[:, :max(Smax, 1)], xl, tl, blank, not fused, _lib.ALGO_AUTO)
        for b in range(B):
            if not np.isfinite(le[b]):
                assert not np.isfinite(la[b]) or np.isnan(la[b]) or la[b] == le[b], (case, b)
                continue
            # (f32 noise floor: a posterior row recomputed over 16 f32 steps carries a few 1e-6 relative error, which is
            # an absolute error of that size where the posterior is ~1 -- logits of scale >= 8; the tests' 2e-6 holds up
            # to scale 5)
            atol = 2e-6 if sharp < 8 else 6e-6
            viol = (np.abs(ga[b].astype(np.float64) - ge[b].astype(np.float64)) - (atol + 1e-4 * np.abs(ge[b].astype(np.float64)))).max()
            if viol > 0 or abs(float(la[b]) - float(le[b])) > 1e-4 * max(1.0, abs(float(le[b]))):
                auto_bad += 1
                print("AUTO case %d utt %d: beyond tolerance by %.2e, loss %.6g vs %.6g (fast path flagged: %s; B=%d T=%d V=%d S=%d sharp=%g fused=%d blank=%d xl=%d tl=%d)" % (
                    case, b, viol, la[b], le[b], bool(np.isnan(lf[b])), B, T, V, Smax, sharp, fused, blank, xl[b], tl[b]))
    for b in range(B):
        total += 1
        if np.isfinite(le[b]): feas[(sharp, "S/T>=.6" if float(tl[b]) / float(xl[b]) >= 0.6 else "S/T<.6")] += 1
        if np.isnan(lf[b]):
            flagged += 1
            key = (sharp, "S/T>=.6" if float(tl[b]) / float(xl[b]) >= 0.6 else "S/T<.6")
            if np.isfinite(le[b]): tally[key] += 1
            if np.isfinite(le[b]) and verbose: print("case %d utt %d: flagged but feasible (B=%d T=%d V=%d S=%d sharp=%g xl=%d tl=%d)" % (case, b, B, T, V, Smax, sharp, xl[b], tl[b]))
            continue
        dl = abs(float(lf[b]) - float(le[b])) / max(1.0, abs(float(le[b])))
        dg = float(np.abs(gf[b].astype(np.float64) - ge[b].astype(np.float64)).max())
        if zd is not None:
            dev = float(np.abs(zd[b, :(int(xl[b]) + 15) // 16]).max())       # worst row of the utterance's segments
            viol_f = float((np.abs(gf[b].astype(np.float64) - ge[b].astype(np.float64)) - (2e-6 + 1e-4 * np.abs(ge[b].astype(np.float64)))).max())
            (zdev_bad if viol_f > 0 else zdev_good).append((dev, viol_f, case, b))
        if dg > 2e-5:
            err = np.abs(gf[b].astype(np.float64) - ge[b].astype(np.float64)).max(-1)      # per frame
            bad = np.nonzero(err > 2e-5)[0]
            print("   frames with error: %d of %d, first %d last %d; by segment: %s" % (len(bad), int(xl[b]), bad[0], bad[-1], sorted(set((bad // 16).tolist()))[:12]))
        if dl > 1e-4 or dg > 2e-5: print("case %d utt %d: loss rel %.2e grad abs %.2e (B=%d T=%d V=%d S=%d sharp=%g fused=%d blank=%d xl=%d tl=%d)" % (case, b, dl, dg, B, T, V, Smax, sharp, fused, blank, xl[b], tl[b]))
        worst_l, worst_g = max(worst_l, dl), max(worst_g, dg)
print("%d utterances in %d cases: %d flagged by the fast path; worst loss rel %.2e, worst grad abs %.2e; AUTO beyond tolerance: %d" % (total, n_cases, flagged, worst_l, worst_g, auto_bad))
for k in sorted(feas): print("  sharp %-4g %-8s feasible %4d, flagged by the fast path %4d" % (k[0], k[1], feas[k], tally[k]))
if zdev_good or zdev_bad:
    gd = np.array([d[0] for d in zdev_good]) if zdev_good else np.zeros(1)
    print("self-check |log2 dev| of utterances within tolerance: median %.2e, 99%% %.2e, max %.2e (%d)" % (np.median(gd), np.quantile(gd, 0.99), gd.max(), len(gd)))
    for d in sorted(zdev_bad): print("   beyond tolerance by %.2e: |log2 dev| %.2e (case %d utt %d)" % (d[1], d[0], d[2], d[3]))
print("flag words of feasible utterances (1 lengths, 2 blank label, 4 infeasible/inf, 8 range, 16 non-finite, 32 log Z mismatch, 64 tiny emissions):")
for k in sorted(flagwords): print("  sharp %-4g flags %3d: %d" % (k[0], k[1], flagwords[k]))
