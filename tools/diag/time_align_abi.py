"""Time e2e_ctc_align through the C ABI (B=64, T=1000, V=29, S in [100,200]); optional argument: another build of the library."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch, ctypes as C
from end2end_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else _lib.LIB_PATH
L = _lib.load()
d = torch.device("cuda", 0)
g = torch.Generator().manual_seed(4)
B, T, V, S = 64, 1000, 29, 200
lp = torch.log_softmax(torch.randn(B, T, V, generator=g), -1).to(d)
tg = torch.randint(1, V, (B, S), generator=g).to(d)
xl = torch.full((B,), T).to(d); tl = torch.randint(100, 201, (B,), generator=g).to(d)
out = torch.empty(B, T, dtype=torch.long, device=d)
n = L.e2e_ctc_align_workspace_bytes(B, T, V, S, 1); ws = torch.empty(n, dtype=torch.uint8, device=d)
def call():
    rc = L.e2e_ctc_align(lp.data_ptr(), 0, *lp.stride(), tg.data_ptr(), tg.stride(0), xl.data_ptr(), tl.data_ptr(), B, T, V, S, 0, 1, out.data_ptr(), -1, ws.data_ptr(), ws.numel(), None)
    assert rc == 0
for _ in range(3): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): call()
e1.record(); torch.cuda.synchronize()
print(os.path.basename(_lib.LIB_PATH), "%.3f ms" % (e0.elapsed_time(e1) / 10))
