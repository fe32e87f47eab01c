"""Every kernel family besides the headline loss, a few calls each at its BASELINE shape (for rocprofv3 --kernel-trace --stats):
greedy (configs[2]), beam 100 without / with the synthetic 3-gram through the fast kernel (configs[3]), the general beam
kernel at V = 8000, the wide-alphabet loss (one GPU's share of configs[4]), the forced alignment."""
import os, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
import bench
from end2end_amd import CTCDecoder, _lib
from end2end_amd.utils.alignment import get_alignment_3d
d = torch.device("cuda", 0)
g = torch.Generator().manual_seed(2)
labels = ["_"] + [chr(97 + i) for i in range(26)] + [" ", "'"]
x = (torch.randn(1024, 1500, 29, generator=g) * 3).to(d)
xl = torch.full((1024,), 1500, dtype=torch.long, device=d)
eng = CTCDecoder(beam_width=1, blank_idx=0, keep_on_device=True)._decoder
for _ in range(5): eng.decode_greedy(x, xl)
xb = torch.log_softmax(x[:64], -1); xlb = xl[:64]
eng = CTCDecoder(beam_width=100, blank_idx=0, after_logsoftmax=True, labels=labels, wip=1.0)._decoder
for _ in range(2): eng.decode(xb, xlb)
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "synthetic_3gram.arpa")
    bench.synthetic_arpa(path, labels)
    eng = CTCDecoder(beam_width=100, blank_idx=0, after_logsoftmax=True, labels=labels, lm_path=path, lmwt=1.0, wip=1.0, oov_penalty=-10.0)._decoder
    for _ in range(2): eng.decode(xb, xlb)
xw = torch.log_softmax(torch.randn(16, 256, 8000, generator=g) * 3, -1).to(d)
eng = CTCDecoder(beam_width=100, blank_idx=0, after_logsoftmax=True)._decoder
for _ in range(2): eng.decode(xw, torch.full((16,), 256, dtype=torch.long, device=d))
del xw
_, wb = bench.make_batch(5000, 512, 256, 8000, 64, d)
wp = bench.HotPath(wb)
for _ in range(4): wp.call()
del wp, wb
lp = torch.log_softmax(torch.randn(64, 1000, 29, generator=g), -1).to(d)
tg = torch.randint(1, 29, (64, 200), generator=g)
for _ in range(3): get_alignment_3d(lp, tg, torch.full((64,), 1000), torch.randint(100, 201, (64,), generator=g))
torch.cuda.synchronize()
print("ok")
