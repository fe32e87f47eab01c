"""Time one loss call shape through the C ABI: time_shape.py B T V S [reps] (diagnostic; use under rocprofv3 for the kernels)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from end2end_amd import _lib
if os.environ.get("E2E_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["E2E_LIB"])      # (an A/B build of the library)
L = _lib.load(); d = torch.device("cuda", 0)
B, T, V, S = (int(a) for a in sys.argv[1:5]); reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
gen = torch.Generator().manual_seed(5)
x = torch.randn(B, T, V, generator=gen).to(d); tg = torch.randint(1, V, (B, S), generator=gen).to(d)
tl = torch.randint(S // 2, S + 1, (B,), generator=gen).to(d); xl = torch.full((B,), T).to(d)
losses = torch.empty(B, device=d); grads = torch.empty(B, T, V, device=d)
n = L.e2e_ctc_loss_workspace_bytes(B, T, V, S, 0, 0); ws = torch.zeros(n, dtype=torch.uint8, device=d)
def call():
    _lib.check(L.e2e_ctc_loss_fwd_bwd(x.data_ptr(), 0, 0, *x.stride(), tg.data_ptr(), tg.stride(0), xl.data_ptr(), tl.data_ptr(),
                                      B, T, V, S, 0, losses.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(), 0, None))
for _ in range(3): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): call()
e1.record(); torch.cuda.synchronize()
print("B=%d T=%d V=%d S<=%d: %.1f us per call" % (B, T, V, S, e0.elapsed_time(e1) / reps * 1e3))
