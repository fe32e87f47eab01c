"""Time the greedy decode kernel at BASELINE configs[2] (B=1024, T=1500, V=29) through the C ABI (HIP events)."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
import end2end_amd._lib as _lib
if os.environ.get('E2E_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['E2E_LIB'])
L = _lib.load(); d = torch.device("cuda", 0)
g = torch.Generator().manual_seed(2)
for (B, T, V) in [(1024, 1500, 29), (1024, 1500, 32), (256, 1000, 29)]:
    x = (torch.randn(B, T, V, generator=g) * 3).to(d)
    xl = torch.full((B,), T, dtype=torch.long, device=d)
    out = torch.empty(B, T, dtype=torch.long, device=d); ol = torch.empty(B, dtype=torch.long, device=d)
    def call():
        rc = L.e2e_ctc_greedy(x.data_ptr(), 0, *x.stride(), xl.data_ptr(), B, T, V, 0, out.data_ptr(), ol.data_ptr(), None)
        assert rc == 0, L.e2e_last_error()
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("B=%d T=%d V=%d: %.1f us, %.2f TB/s algorithmic" % (B, T, V, us, B * T * (V * 4 + 8) / us / 1e6))
