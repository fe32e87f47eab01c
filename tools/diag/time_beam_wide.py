"""Time prefix beam search on alphabets beyond the one-workgroup-LDS kernel (general kernel), and the headline decode
shape through both kernels (E2E_BEAM_GENERAL=1 forces the general one)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from end2end_amd import CTCDecoder
d = torch.device("cuda", 0)
g = torch.Generator().manual_seed(2)
for (B, T, V, W) in [(64, 1500, 29, 100), (64, 300, 200, 100), (64, 256, 1000, 100), (16, 256, 8000, 100)]:
    x = torch.log_softmax((torch.randn(B, T, V, generator=g) * 3), -1).to(d)
    xl = torch.full((B,), T, dtype=torch.long, device=d)
    eng = CTCDecoder(beam_width=W, blank_idx=0, after_logsoftmax=True)._decoder
    eng.decode(x, xl); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = eng.decode(x, xl); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("general=%s B=%d T=%d V=%d W=%d: %.1f ms, %.1f us/step, %.0f utt/s" % (os.environ.get("E2E_BEAM_GENERAL", "0"), B, T, V, W, dt * 1e3, dt * 1e6 / T, B / dt))
