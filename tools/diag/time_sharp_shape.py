"""One loss shape at one logit scale against unrelated targets (the cliff regimes): time_sharp_shape.py B T V S SCALE"""
import ctypes, sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
from end2end_amd import _lib
L = _lib.load(); d = torch.device("cuda", 0)
B, T, V, S = (int(a) for a in sys.argv[1:5]); sharp = float(sys.argv[5])
gen = torch.Generator().manual_seed(3)
x = (torch.randn(B, T, V, generator=gen) * sharp).to(d); tg = torch.randint(1, V, (B, S), generator=gen).to(d)
tl = torch.randint(S // 2, S + 1, (B,), generator=gen).to(d); xl = torch.full((B,), T).to(d)
losses = torch.empty(B, device=d); grads = torch.empty(B, T, V, device=d)
n = L.e2e_ctc_loss_workspace_bytes(B, T, V, S, 0, 0); ws = torch.zeros(n, dtype=torch.uint8, device=d)
def call():
    _lib.check(L.e2e_ctc_loss_fwd_bwd(x.data_ptr(), 0, 0, *x.stride(), tg.data_ptr(), tg.stride(0), xl.data_ptr(), tl.data_ptr(),
                                      B, T, V, S, 0, losses.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(), 0, None))
call(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): call()
e1.record(); torch.cuda.synchronize()
fl = (ctypes.c_int * B)(); lz = (ctypes.c_double * (2 * B))(); un = ctypes.c_int(0)
L.e2e_debug_fast_state.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
L.e2e_debug_fast_redo_failures.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
L.e2e_debug_fast_state(ws.data_ptr(), B, T, V, S, fl, lz); L.e2e_debug_fast_redo_failures(ws.data_ptr(), B, T, V, S, ctypes.byref(un))
print("B=%d T=%d V=%d S<=%d scale %g: %.3f ms per call, %d of %d utterances left the f32 lattice, %d recomputed by the exact kernel, finite %s" % (
    B, T, V, S, sharp, e0.elapsed_time(e1) / 3, sum(1 for v in fl if v & 511), B, un.value, bool(torch.isfinite(losses).all())))
