"""The f64 redo of the segment kernel (flag bits 8/16): E2E_ALGO_AUTO on batches where the f32 segment kernel gives up must
agree with the exact kernel, and cost far less than the full exact fallback used to."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import gpu_util as U
from end2end_amd import _lib
rng = np.random.default_rng(0)
for (B, T, V, S, boost, bunch) in [(32, 1000, 29, 200, 8.0, 0.6), (16, 500, 40, 120, 12.0, 0.5), (8, 300, 20, 60, 16.0, 0.4), (8, 700, 64, 255, 8.0, 0.7)]:
    x = rng.standard_normal((B, T, V)).astype(np.float32)
    tg = rng.integers(1, V, size=(B, S)); tl = rng.integers(S // 2, S + 1, size=B); xl = np.full(B, T); xl[1:] -= rng.integers(0, 40, size=B - 1)
    for b in range(B):
        Lb = int(tl[b]); slots = np.sort(rng.choice(np.arange(0, int(xl[b] * bunch)), size=Lb, replace=False))
        x[b, slots, tg[b, :Lb]] += boost
    xt, tgt, xlt, tlt = torch.from_numpy(x), torch.from_numpy(tg), torch.from_numpy(xl), torch.from_numpy(tl)
    lf, gf = U.c_abi_loss(xt, tgt, xlt, tlt, 0, False, _lib.ALGO_FAST)
    la, ga = U.c_abi_loss(xt, tgt, xlt, tlt, 0, False, _lib.ALGO_AUTO)
    le, ge = U.c_abi_loss(xt, tgt, xlt, tlt, 0, False, _lib.ALGO_EXACT)
    fl = np.isnan(lf)
    dl = np.abs(la - le) / np.maximum(1, np.abs(le)); dg = np.abs(ga.astype(np.float64) - ge.astype(np.float64)).reshape(B, -1).max(1)
    viol = (np.abs(ga.astype(np.float64) - ge.astype(np.float64)) - (2e-6 + 1e-4 * np.abs(ge.astype(np.float64)))).max()
    print("B=%d T=%d V=%d S<=%d boost %g: %d flagged by the f32 path; auto vs exact: loss rel %.1e, grad abs %.1e (flagged ones: %.1e); beyond rtol 1e-4 / atol 2e-6 by %.1e" % (
        B, T, V, S, boost, int(fl.sum()), dl.max(), dg.max(), dg[fl].max() if fl.any() else 0.0, max(viol, 0.0)))
