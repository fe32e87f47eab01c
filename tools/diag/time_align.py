"""Time the forced-alignment kernel at B=64, T=1000, V=29, S in [100,200] (CTC) through the Python helper."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from end2end_amd.utils.alignment import get_alignment_3d
d = torch.device("cuda", 0)
g = torch.Generator().manual_seed(4)
lp = torch.log_softmax(torch.randn(64, 1000, 29, generator=g), -1).to(d)
tg = torch.randint(1, 29, (64, 200), generator=g).to(d)
xl = torch.full((64,), 1000).to(d); tl = torch.randint(100, 201, (64,), generator=g).to(d)
for _ in range(3): get_alignment_3d(lp, tg, xl, tl)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): get_alignment_3d(lp, tg, xl, tl)
e1.record(); torch.cuda.synchronize()
print("forced alignment B=64 T=1000 S<=200: %.3f ms per call (%.0f utt/s)" % (e0.elapsed_time(e1) / 10, 64 / (e0.elapsed_time(e1) / 10) * 1e3))
