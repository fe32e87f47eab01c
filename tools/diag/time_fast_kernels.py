"""Time the fast path's two kernels for a few (T, V, S) shapes with HIP events (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from end2end_amd import _lib
if os.environ.get('E2E_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['E2E_LIB'])
L = _lib.load()
d = torch.device("cuda", 0)
def run(B, T, V, S, reps=20):
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, V, generator=gen).to(d); tg = torch.randint(1, V, (B, S), generator=gen).to(d)
    tl = torch.randint(max(S // 2, 1), S + 1, (B,), generator=gen).to(d); xl = torch.full((B,), T).to(d)
    losses = torch.empty(B, device=d); grads = torch.empty(B, T, V, device=d)
    n = L.e2e_ctc_loss_workspace_bytes(B, T, V, S, 0, 2); ws = torch.zeros(n, dtype=torch.uint8, device=d)
    def call():
        rc = L.e2e_ctc_loss_fwd_bwd(x.data_ptr(), 0, 0, *x.stride(), tg.data_ptr(), tg.stride(0), xl.data_ptr(), tl.data_ptr(),
                                    B, T, V, S, 0, losses.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(), 2, None)
        assert rc == 0, L.e2e_last_error()
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nan = int(torch.isnan(losses).sum())
    print("B=%d T=%d V=%d S<=%d: %.1f us per call, %d flagged" % (B, T, V, S, ms * 1e3, nan))
for shape in [(256, 1000, 29, 200), (256, 1000, 29, 127), (256, 1000, 29, 63), (256, 500, 29, 100), (512, 1000, 29, 200), (1024, 1000, 29, 200), (256, 1000, 64, 200)]:
    run(*shape)
print("wide alphabet (C5 per-GPU share):")
run(512, 256, 8000, 64, reps=5)
