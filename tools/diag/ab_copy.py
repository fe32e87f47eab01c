import ctypes as C, os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
from end2end_amd import _lib
d = torch.device("cuda", 0)
n = 1 << 30
src = torch.empty(n, dtype=torch.uint8, device=d).zero_(); dst = torch.empty_like(src)
libs = {}
for pth in sys.argv[1:]:
    L = C.CDLL(os.path.abspath(pth)); L.e2e_debug_stream_copy.restype = C.c_int; L.e2e_debug_stream_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]; libs[os.path.basename(pth)] = L
res = {k: [] for k in libs}
for rnd in range(8):
    for k, L in libs.items():
        for _ in range(2): L.e2e_debug_stream_copy(dst.data_ptr(), src.data_ptr(), n, None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): L.e2e_debug_stream_copy(dst.data_ptr(), src.data_ptr(), n, None)
        e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 5)
for k, v in res.items(): print("%-24s %.2f TB/s" % (k, 2 * n / (statistics.median(v) * 1e-3) / 1e12))
