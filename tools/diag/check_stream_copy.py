import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from end2end_amd import _lib
d = torch.device("cuda", 0); L = _lib.load()
for n in (1 << 20, (1 << 20) + 4 * 777, 4 * 5):
    src = torch.randn(n, device=d); dst = torch.zeros_like(src)
    _lib.check(L.e2e_debug_stream_copy(dst.data_ptr(), src.data_ptr(), n * 4, _lib.stream_ptr(d)))
    torch.cuda.synchronize(); assert torch.equal(src, dst), n
print("copy ok; peak_measured %.0f GB/s" % bench.measured_copy_gbs(torch, d))
