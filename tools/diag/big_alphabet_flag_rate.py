"""How many utterances does the wide-row form of the fast path hand over (ALGO_FAST: NaN) on random logits at wide alphabets?"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import gpu_util as U
from end2end_amd import _lib
for (B, T, V, S, sharp) in [(64, 600, 448, 100, 1.0), (64, 600, 448, 100, 0.1), (64, 400, 300, 150, 1.0), (64, 256, 200, 150, 1.0), (64, 256, 200, 150, 0.1), (32, 700, 400, 400, 1.0)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, V, generator=g) * sharp
    tg = torch.randint(1, V, (B, S), generator=g)
    xl = torch.randint(T // 2, T + 1, (B,), generator=g); tl = torch.randint(S // 2, S + 1, (B,), generator=g)
    tl = torch.minimum(tl, xl // 2)
    lf, gf = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_FAST)
    print("B%d T%d V%d S%d sharp %.1f: flagged %d of %d" % (B, T, V, S, sharp, int(np.isnan(lf).sum()), B), flush=True)
# the flag words of the first shape (1 lengths, 2 blank label, 4 infeasible / inf, 8 range, 16 non-finite, 32 log Z mismatch, 64 tiny
# emissions, 128 hand-off timeout)
import ctypes, collections
B, T, V, S = 64, 600, 448, 100
g = torch.Generator().manual_seed(1)
x = torch.randn(B, T, V, generator=g)
tg = torch.randint(1, V, (B, S), generator=g)
xl = torch.randint(T // 2, T + 1, (B,), generator=g); tl = torch.minimum(torch.randint(S // 2, S + 1, (B,), generator=g), xl // 2)
kept = {}
lf, gf = U.c_abi_loss(x, tg, xl, tl, 0, False, _lib.ALGO_FAST, keep=kept)
fw = (ctypes.c_int * B)(); lz = (ctypes.c_double * (2 * B))()
L = _lib.load(); L.e2e_debug_fast_state.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
assert L.e2e_debug_fast_state(kept["workspace"].data_ptr(), B, T, V, S, fw, lz) == 0
print("flag words:", collections.Counter(int(f) for f in fw), "; flagged (T, S):", [(int(xl[b]), int(tl[b])) for b in range(B) if fw[b]])
