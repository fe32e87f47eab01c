"""Calibration data for a detector of utterances the packed-f32 chains (E2E_CHAINS_F32) get wrong beyond the DEFAULT tolerance
(2e-6 + 1e-4 |g|): per utterance the fast path keeps, |log Z alpha-side - log Z beta-side| of the f32 chains against the worst
excess of its gradient over the tolerance (exact kernel as the truth).  Regimes: noise of scale 0.1 .. 5 against unrelated
targets, peaky emissions consistent with the targets, the same with mislabelled utterances; T 400 .. 2000, 128 .. 255 labels.
  python tools/diag/f32_detector_calib.py [seed]"""
import ctypes, sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import gpu_util as U
from end2end_amd import _lib
L = _lib.load()
L.e2e_debug_fast_state.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
HAVE_ZDEV = hasattr(L, "e2e_debug_fast_zdev")          # (the instrumented build: E2E_CTC_LIB=build/diag/prof_lib.so)
if HAVE_ZDEV: L.e2e_debug_fast_zdev.argtypes = [ctypes.c_void_p, ctypes.c_int]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
rows = []
def run(mode, x, tg, xl, tl, V):
    B, T = x.shape[0], x.shape[1]; S = tg.shape[1]
    le, ge = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_EXACT)
    keep = {}
    zd = (ctypes.c_float * 16384)()
    if HAVE_ZDEV: L.e2e_debug_fast_zdev(zd, 1)
    lf, gf = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_FAST, keep=keep, chains=_lib.CHAINS_F32)
    if HAVE_ZDEV: L.e2e_debug_fast_zdev(zd, 0)
    NS = (T + 15) // 16
    zd = np.frombuffer(zd, dtype=np.float32)[:B * NS].reshape(B, NS) if B * NS <= 16384 else np.zeros((B, NS), np.float32)
    fl = (ctypes.c_int * B)(); lz = (ctypes.c_double * (2 * B))()
    L.e2e_debug_fast_state(keep["workspace"].data_ptr(), B, T, V, S, fl, lz)
    for b in range(B):
        if np.isnan(lf[b]) or not np.isfinite(le[b]): continue
        d = np.abs(gf[b].astype(np.float64) - ge[b]); tol = 2e-6 + 1e-4 * np.abs(ge[b])
        rows.append((mode, T, int(tl[b]), abs(lz[2 * b] - lz[2 * b + 1]), float((d - tol).max()), float(d.max()), abs(float(lf[b]) - float(le[b])) / max(1.0, abs(float(le[b]))), float(le[b]),
                     float(zd[b].max()), float(zd[b].mean())))
for rep in range(6):
    for T in (400, 1000, 2000):
        V = int(rng.choice([29, 48])); B = 8; S = int(rng.integers(140, min(255, T // 2) + 1))
        tg = torch.tensor(rng.integers(1, V, size=(B, S)), dtype=torch.long); tl = torch.tensor(rng.integers(128, S + 1, size=B)); xl = torch.full((B,), T)
        for scale in (0.1, 1.0, 2.0, 3.0, 5.0):
            x = torch.from_numpy((rng.standard_normal((B, T, V)) * scale).astype(np.float32))
            run("noise x%g" % scale, x, tg, xl, tl, V)
        for boost in (2.0, 6.0, 10.0, 14.0, 20.0):
            x = rng.standard_normal((B, T, V)).astype(np.float32)
            for b in range(B):
                src = b if b % 4 else (b + 1) % B                  # every fourth utterance emits another one's transcript
                n = int(tl[src]); slots = np.sort(rng.choice(T, size=n, replace=False)); path = np.zeros(T, dtype=np.int64); path[slots] = tg[src, :n].numpy()
                x[b, np.arange(T), path] += boost
            run("peaky +%g (1 in 4 mislabelled)" % boost, torch.from_numpy(x), tg, xl, tl, V)
rows.sort(key=lambda r: r[3])
print("%-34s %5s %4s %10s %11s %10s %9s %9s %10s %10s" % ("mode", "T", "S", "|dlogZ|", "excess", "max|dg|", "rel dloss", "loss", "max|dev|", "mean|dev|"))
for r in rows: print("%-34s %5d %4d %10.3e %11.3e %10.3e %9.1e %9.1f %10.3e %10.3e" % r)
over = [r for r in rows if r[4] > 0]
print("%d kept utterances, %d over the default tolerance; smallest |dlogZ| among those over: %s; largest |dlogZ| among those within: %.3e" % (
    len(rows), len(over), ("%.3e" % min(r[3] for r in over)) if over else "-", max(r[3] for r in rows if r[4] <= 0)))
