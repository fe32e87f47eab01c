"""One GPU's share of BASELINE configs[4] (B=512, T=256, V=8000, S<=64): run the C-ABI call a few times (for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from end2end_amd import _lib
L = _lib.load(); d = torch.device("cuda", 0)
B, T, V, S = 512, 256, 8000, 64
gen = torch.Generator().manual_seed(0)
DT = {"bf16": torch.bfloat16, "f16": torch.float16}.get(sys.argv[1] if len(sys.argv) > 1 else "", torch.float32)
x = torch.randn(B, T, V, generator=gen).to(d).to(DT); tg = torch.randint(1, V, (B, S), generator=gen).to(d)
tl = torch.randint(S // 2, S + 1, (B,), generator=gen).to(d); xl = torch.full((B,), T).to(d)
code = _lib.dtype_code(DT)
losses = torch.empty(B, device=d); grads = torch.empty(B, T, V, device=d, dtype=DT)
n = L.e2e_ctc_loss_workspace_bytes(B, T, V, S, code, 0); ws = torch.zeros(n, dtype=torch.uint8, device=d)
for _ in range(6):
    rc = L.e2e_ctc_loss_fwd_bwd(x.data_ptr(), code, 0, *x.stride(), tg.data_ptr(), tg.stride(0), xl.data_ptr(), tl.data_ptr(),
                                B, T, V, S, 0, losses.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(), 0, None)
    assert rc == 0
torch.cuda.synchronize()
print("ok", float(losses.mean()))
