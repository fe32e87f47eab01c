"""Which inputs make an f64 redo of a single segment fail with nothing flagged for the chains -- the flagged launch's round 1 (flag word
2568 = 2048 settled in extended range + 512 a segment redo failed + 8 range): tests/test_gpu_regimes.py pins one of them."""
import sys, os, ctypes
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import gpu_util as U
from end2end_amd import _lib
L = _lib.load()
L.e2e_debug_flagged_counters.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
L.e2e_debug_fast_state.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
for (B, T, V, S, scale) in [(3, 2000, 29, 400, 3.0), (8, 2000, 29, 400, 3.0), (8, 2000, 29, 400, 2.5), (8, 2000, 29, 400, 3.5), (8, 1500, 29, 300, 3.0), (8, 1000, 29, 200, 4.0), (8, 1000, 29, 200, 5.0), (8, 2000, 40, 400, 3.0)]:
    for seed in range(4):
        rng = np.random.default_rng(seed)
        x = (rng.standard_normal((B, T, V)) * scale).astype(np.float32)
        tg = rng.integers(1, V, size=(B, S)); tl = rng.integers(S // 2, S + 1, size=B); xl = np.full(B, T)
        keep = {}
        la, ga = U.c_abi_loss(torch.from_numpy(x), tg, xl, tl, 0, False, _lib.ALGO_AUTO, keep=keep)
        to, fr = ctypes.c_int(-1), ctypes.c_int(-1)
        L.e2e_debug_flagged_counters(keep["workspace"].data_ptr(), B, T, V, S, ctypes.byref(to), ctypes.byref(fr))
        fl = (ctypes.c_int * B)(); lz = (ctypes.c_double * (2 * B))()
        L.e2e_debug_fast_state(keep["workspace"].data_ptr(), B, T, V, S, fl, lz)
        print(B, T, V, S, scale, "seed", seed, "failed_redos", fr.value, "timeouts", to.value, "flags", list(fl))
