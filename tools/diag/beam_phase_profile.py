import sys, os, ctypes as C
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
import end2end_amd._lib as _lib
_lib.LIB_PATH = os.path.join(root, "build/diag/prof_lib.so")
from end2end_amd import CTCDecoder
d = torch.device("cuda", 0)
labels = ["_"] + [chr(97 + i) for i in range(26)] + [" ", "'"]
g = torch.Generator().manual_seed(2)
B, T, W = 64, 300, int(sys.argv[1]) if len(sys.argv) > 1 else 100
x = torch.log_softmax((torch.randn(B, T, 29, generator=g) * 3), -1).to(d)
xl = torch.full((B,), T, dtype=torch.long, device=d)
if len(sys.argv) > 2 and sys.argv[2] == "lm":          # the synthetic 3-gram model of bench.py
    import tempfile, bench
    td = tempfile.mkdtemp(); path = os.path.join(td, "synthetic_3gram.arpa"); bench.synthetic_arpa(path, labels)
    eng = CTCDecoder(beam_width=W, blank_idx=0, after_logsoftmax=True, labels=labels, lm_path=path, lmwt=1.0, wip=1.0, oov_penalty=-10.0)._decoder
else:
    eng = CTCDecoder(beam_width=W, blank_idx=0, after_logsoftmax=True, labels=labels, wip=1.0)._decoder
eng.decode(x, xl); eng.decode(x, xl)
buf = (C.c_ulonglong * 16)()
L = _lib.load(); L.e2e_debug_beam_profile.argtypes = [C.c_void_p]
assert L.e2e_debug_beam_profile(buf) == 0
names = ["(unused)", "pairs", "members", "select: rank + place", "(unused)", "guards+tables / LM followers", "(unused)", "select: radix passes", "select: gather", "(unused)", "(passes)", "guards (LM kernel)", "LM state signatures", "LM leaders", "LM rows"]
tot = sum(buf[k] for k in range(15) if k != 10)
for k, nm in enumerate(names):
    if k != 10 and buf[k]: print("%-16s %8.0f cycles/step (%4.1f%%)" % (nm, buf[k] / T, 100.0 * buf[k] / tot))
print("total %.0f cycles/step; radix passes per step %.2f" % (tot / T, buf[10] / T))
