"""Time the exact kernel (E2E_ALGO_EXACT) and the fallback path for a few flagged utterances."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
import end2end_amd._lib as _lib
if os.environ.get('E2E_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['E2E_LIB'])
L = _lib.load(); d = torch.device("cuda", 0)
for (B, T, V, S, dt) in [(256, 1000, 29, 200, torch.float32), (32, 1000, 29, 200, torch.float32), (64, 300, 64, 100, torch.float64)]:
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, V, generator=gen, dtype=torch.float64).to(dt).to(d); tg = torch.randint(1, V, (B, S), generator=gen).to(d)
    tl = torch.randint(S // 2, S + 1, (B,), generator=gen).to(d); xl = torch.full((B,), T).to(d)
    losses = torch.empty(B, device=d, dtype=dt); grads = torch.empty(B, T, V, device=d, dtype=dt)
    code = 0 if dt == torch.float32 else 1
    n = L.e2e_ctc_loss_workspace_bytes(B, T, V, S, code, 1); ws = torch.zeros(n, dtype=torch.uint8, device=d)
    def call():
        rc = L.e2e_ctc_loss_fwd_bwd(x.data_ptr(), code, 0, *x.stride(), tg.data_ptr(), tg.stride(0), xl.data_ptr(), tl.data_ptr(),
                                    B, T, V, S, 0, losses.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(), 1, None)
        assert rc == 0, L.e2e_last_error()
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); call(); call(); e1.record(); torch.cuda.synchronize()
    print("exact kernel B=%d T=%d V=%d S<=%d %s: %.2f ms" % (B, T, V, S, str(dt).split('.')[-1], e0.elapsed_time(e1) / 2))
