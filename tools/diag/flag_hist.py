"""Histogram of the fast path's flag words for a synthetic shape: python3 tools/diag/flag_hist.py B T V S (diagnostic)."""
import sys, os, ctypes as C, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from end2end_amd import _lib
if os.environ.get('E2E_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['E2E_LIB'])
L = _lib.load()
d = torch.device("cuda", 0)
B, T, V, S = [int(a) for a in sys.argv[1:5]]
gen = torch.Generator().manual_seed(0)
x = torch.randn(B, T, V, generator=gen).to(d); tg = torch.randint(1, V, (B, S), generator=gen).to(d)
tl = torch.randint(max(S // 2, 1), S + 1, (B,), generator=gen).to(d); xl = torch.full((B,), T).to(d)
losses = torch.empty(B, device=d); grads = torch.empty(B, T, V, device=d)
n = L.e2e_ctc_loss_workspace_bytes(B, T, V, S, 0, 2); ws = torch.zeros(n, dtype=torch.uint8, device=d)
rc = L.e2e_ctc_loss_fwd_bwd(x.data_ptr(), 0, 0, *x.stride(), tg.data_ptr(), tg.stride(0), xl.data_ptr(), tl.data_ptr(), B, T, V, S, 0,
                            losses.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(), 2, None)
assert rc == 0
fl = (C.c_int * B)(); lz = (C.c_double * (2 * B))()
L.e2e_debug_fast_state.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
L.e2e_debug_fast_state(ws.data_ptr(), B, T, V, S, fl, lz)
fl = np.array(fl[:]); lz = np.array(lz[:]).reshape(B, 2)
print("flag words (1 lengths, 2 blank label, 4 infeasible/inf, 8 self-check/range, 16 non-finite, 32 log Z mismatch, 64 tiny emissions, 128 protocol):")
print(collections.Counter(fl.tolist()))
bad = np.nonzero(fl)[0][:6]
for i in bad: print("  utt %d: S=%d flags %d logZ alpha %.9g beta %.9g" % (i, int(tl[i]), fl[i], lz[i, 0], lz[i, 1]))
