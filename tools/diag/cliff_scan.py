"""Where does a loss call cost far more than its shape's usual time?  Every shape at logits of scale 1, 3 and 8 against
unrelated targets (the fallback regimes), ms per call and the ratio to scale 1."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
from end2end_amd import _lib
L = _lib.load(); d = torch.device("cuda", 0)
shapes = [(256, 1000, 29, 200), (256, 500, 29, 100), (256, 1000, 80, 200), (256, 1000, 29, 120), (64, 1000, 29, 60), (256, 2000, 29, 250),
          (128, 1000, 29, 447), (64, 256, 8000, 200), (256, 256, 8000, 64), (256, 1000, 150, 150), (128, 800, 300, 200), (32, 700, 8000, 400)]
for (B, T, V, S) in shapes:
    base = None
    for sharp in (1.0, 3.0, 8.0):
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
        ms = e0.elapsed_time(e1) / 3
        base = base or ms
        print("B=%-3d T=%-4d V=%-5d S<=%-3d scale %g: %8.3f ms  (x%.1f)%s" % (B, T, V, S, sharp, ms, ms / base, "   <-- cliff" if ms / base > 10 else ""), flush=True)
        del x, grads, ws
        torch.cuda.empty_cache()
