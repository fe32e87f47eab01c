"""Time the wide-alphabet call (one GPU's share of BASELINE configs[4]) through the C ABI (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from end2end_amd import _lib
L = _lib.load(); d = torch.device("cuda", 0)
B, T, V, S = 512, 256, 8000, 64
gen = torch.Generator().manual_seed(5)
dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "f32"]
code = _lib.dtype_code(dt); esz = 2 if code else 4
x = torch.randn(B, T, V, generator=gen).to(dt).to(d); tg = torch.randint(1, V, (B, S), generator=gen).to(d)
tl = torch.randint(S // 2, S + 1, (B,), generator=gen).to(d); xl = torch.full((B,), T).to(d)
losses = torch.empty(B, device=d); grads = torch.empty(B, T, V, device=d, dtype=dt)
n = L.e2e_ctc_loss_workspace_bytes(B, T, V, S, code, 0); ws = torch.zeros(n, dtype=torch.uint8, device=d)
def call():
    rc = L.e2e_ctc_loss_fwd_bwd(x.data_ptr(), code, 0, *x.stride(), tg.data_ptr(), tg.stride(0), xl.data_ptr(), tl.data_ptr(),
                                B, T, V, S, 0, losses.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(), 0, None)
    assert rc == 0, L.e2e_last_error()
for _ in range(3): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): call()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("wide B=512 T=256 V=8000 %s: %.1f us per call, %.3f of 8 TB/s; loss0 %.4f gsum %.6f" % (dt, ms * 1e3, 2.0 * V * esz * B * T / (ms * 1e-3) / 8e12, losses[0].item(), grads[3].abs().sum().item()))
