import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import end2end_amd._lib as _lib
if os.environ.get('E2E_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['E2E_LIB'])
from end2end_amd import CTCDecoder
d = torch.device("cuda", 0)
labels = ["_"] + [chr(97 + i) for i in range(26)] + [" ", "'"]
g = torch.Generator().manual_seed(2)
for (B, T, W) in [(64, 300, 100), (64, 1500, 100), (64, 300, 20)]:
    x = torch.log_softmax((torch.randn(B, T, 29, generator=g) * 3), -1).to(d)
    xl = torch.full((B,), T, dtype=torch.long, device=d)
    eng = CTCDecoder(beam_width=W, blank_idx=0, after_logsoftmax=True, labels=labels, wip=1.0)._decoder
    eng.decode(x, xl); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = eng.decode(x, xl); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("B=%d T=%d W=%d: %.1f ms, %.1f us/step, %.0f utt/s  [%s...]" % (B, T, W, dt * 1e3, dt * 1e6 / T, B / dt, r[2][0][:30]))
