"""Per-phase cycle sums of the fast path's segment kernel (instrumented build build/diag/prof_lib.so)."""
import sys, ctypes as C, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np, torch
import end2end_amd._lib as _lib
_lib.LIB_PATH = os.path.join(root, "build/diag/prof_lib.so")
L = _lib.load()
d = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(0)
B, T, V, S = 256, 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 29, int(sys.argv[1]) if len(sys.argv) > 1 else 200      # argv: S [V]
x = torch.randn(B, T, V, generator=gen).to(d); tg = torch.randint(1, V, (B, S), generator=gen).to(d)
tl = torch.randint(S // 2, S + 1, (B,), generator=gen).to(d); xl = torch.full((B,), T).to(d)
losses = torch.empty(B, device=d); grads = torch.empty(B, T, V, device=d)
n = L.e2e_ctc_loss_workspace_bytes(B, T, V, S, 0, 2); ws = torch.zeros(n, dtype=torch.uint8, device=d)
def call():
    rc = L.e2e_ctc_loss_fwd_bwd(x.data_ptr(), 0, 0, *x.stride(), tg.data_ptr(), tg.stride(0), xl.data_ptr(), tl.data_ptr(), B, T, V, S, 0,
                                losses.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(), 2, None)
    assert rc == 0
for _ in range(3): call()
L.e2e_debug_fast_profile2.argtypes = [C.c_void_p, C.c_int]
assert L.e2e_debug_fast_profile2(None, 1) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); call(); e1.record(); torch.cuda.synchronize()
print('instrumented call: %.1f us' % (e0.elapsed_time(e1) * 1e3))
buf = (C.c_ulonglong * (16384 * 8))()
assert L.e2e_debug_fast_profile2(buf, 0) == 0
A = np.frombuffer(buf, dtype=np.uint64).astype(np.float64).reshape(16384, 8)
live = A.sum(1) > 0
nseg = live.sum(); a = A[live].sum(0)
names = ["prologue+staging", "alpha ckpt load", "alpha pass", "beta ckpt load", "beta half-pass (x2)", "scan rows (x2)", "gradient rows (x2)"]
tot = a[:7].sum()
for i, nm in enumerate(names):
    print("%-22s %8.0f cycles per segment  (%4.1f%%)" % (nm, a[i] / nseg, 100 * a[i] / tot))
print("total %.0f cycles per segment" % (tot / nseg))
