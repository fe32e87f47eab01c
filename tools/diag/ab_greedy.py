"""A/B builds of libe2e_ctc.so on the greedy decode in ONE process (same buffers, interleaved rounds, median):
   python tools/diag/ab_greedy.py lib1.so lib2.so ...      AB_B / AB_T / AB_V: shape (default BASELINE configs[2]); AB_DTYPE=bf16 / f16 / f64"""
import ctypes as C, os, sys, statistics
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
from end2end_amd import _lib
def bind(path):
    L = C.CDLL(path); L.e2e_ctc_greedy.restype = C.c_int; L.e2e_ctc_greedy.argtypes = _lib.load().e2e_ctc_greedy.argtypes; return L
libs = {os.path.basename(p): bind(os.path.join(root, p)) for p in sys.argv[1:]}
d = torch.device("cuda", 0)
B, T, V = (int(os.environ.get(k, v)) for k, v in (("AB_B", "1024"), ("AB_T", "1500"), ("AB_V", "29")))
DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f64": torch.float64}.get(os.environ.get("AB_DTYPE", ""), torch.float32)
g = torch.Generator().manual_seed(2)
x = (torch.randn(B, T, V, generator=g) * 3).to(DT).to(d); xl = torch.full((B,), T, dtype=torch.long, device=d)
out = torch.empty(B, T, dtype=torch.long, device=d); ol = torch.empty(B, dtype=torch.long, device=d)
def call(L):
    assert L.e2e_ctc_greedy(x.data_ptr(), _lib.dtype_code(DT), *x.stride(), xl.data_ptr(), B, T, V, 0, out.data_ptr(), ol.data_ptr(), None) == 0
res = {k: [] for k in libs}; ref = None
for rnd in range(12):
    for k, L in libs.items():
        for _ in range(3): call(L)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call(L)
        e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 20 * 1e3)
        if rnd == 0:
            cur = (out.clone(), ol.clone())
            if ref is None: ref = cur
            else: assert torch.equal(ref[0], cur[0]) and torch.equal(ref[1], cur[1]), "results differ between the builds"
bytes_ = B * T * (V * x.element_size() + 8)
for k, v in res.items():
    print("%-28s median %.1f us  min %.1f us  (%.2f TB/s algorithmic)" % (k, statistics.median(v), min(v), bytes_ / statistics.median(v) / 1e6))
