import sys, os
sys.path.insert(0, os.getcwd())
import torch
from end2end_amd import _lib
L = _lib.load()
d = torch.device("cuda", 0)
def run(B, T, V, S, lo, reps=20):
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, V, generator=gen).to(d); tg = torch.randint(1, V, (B, S), generator=gen).to(d)
    tl = torch.randint(lo, S + 1, (B,), generator=gen).to(d); xl = torch.full((B,), T).to(d)
    losses = torch.empty(B, device=d); grads = torch.empty(B, T, V, device=d)
    n = L.e2e_ctc_loss_workspace_bytes(B, T, V, S, 0, 2); ws = torch.zeros(n, dtype=torch.uint8, device=d)
    def call():
        rc = L.e2e_ctc_loss_fwd_bwd(x.data_ptr(), 0, 0, *x.stride(), tg.data_ptr(), tg.stride(0), xl.data_ptr(), tl.data_ptr(),
                                    B, T, V, S, 0, losses.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(), 2, None)
        assert rc == 0
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    print("B=%d T=%d V=%d S in [%d,%d]: %.1f us per call, %d flagged" % (B, T, V, lo, S, e0.elapsed_time(e1) / reps * 1e3, int(torch.isnan(losses).sum())))
run(256, 1000, 29, 255, 200); run(256, 1000, 29, 255, 128); run(256, 1000, 29, 200, 100); run(256, 1000, 29, 223, 112)
run(256, 1000, 29, 300, 256, reps=5); run(256, 2000, 29, 400, 300, reps=5); run(64, 2000, 29, 447, 400, reps=5)
