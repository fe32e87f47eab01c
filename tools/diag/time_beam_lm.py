"""Time beam 100 with the synthetic 3-gram at BASELINE configs[3] (B=64, T=1500, V=29); E2E_LM_TWO_ROUNDS=1: the id-keyed lookup."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from end2end_amd import CTCDecoder
d = torch.device("cuda", 0)
labels = ["_"] + [chr(97 + i) for i in range(26)] + [" ", "'"]
g = torch.Generator().manual_seed(2)
x = torch.log_softmax((torch.randn(64, 1500, 29, generator=g) * 3), -1).to(d)
xl = torch.full((64,), 1500, dtype=torch.long, device=d)
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "synthetic_3gram.arpa")
    bench.synthetic_arpa(path, labels)
    eng = CTCDecoder(beam_width=100, blank_idx=0, after_logsoftmax=True, labels=labels, lm_path=path, lmwt=1.0, wip=1.0, oov_penalty=-10.0)._decoder
    eng.decode(x, xl); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); r = eng.decode(x, xl); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    dt = min(ts)
    print("two_rounds=%s: %.1f ms, %.0f utt/s [%s]" % (os.environ.get("E2E_LM_TWO_ROUNDS", "0"), dt * 1e3, 64 / dt, r[2][0][:40]))
