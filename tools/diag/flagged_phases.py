"""Where the flagged-utterance launch spends its time: bench.py's emission regimes (and the fallback regime) with the phase stamps
of e2e_debug_flagged_phases (workgroup 0's view, microseconds since the launch's start)."""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
import bench
from end2end_amd import _lib
L = _lib.load(); dev = torch.device("cuda", 0)
L.e2e_debug_flagged_phases.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
w = bench.WORKLOAD; B, T, V, S = w["B"], w["T"], w["V"], w["S"]
def run(name, host):
    hp = bench.HotPath(tuple(t.to(dev) for t in host))
    for _ in range(3): hp.call(hp.means[0, :1])
    ms = bench.time_events(torch, lambda: hp.call(hp.means[0, :1]), 5)
    us = (ctypes.c_double * 6)()
    L.e2e_debug_flagged_phases(hp.ws.data_ptr(), B, T, V, S, us)
    print("%-16s %.3f ms per call; flagged launch: redo %.0f, wait %.0f, chains %.0f, segments %.0f, end %.0f us (last workgroup known at %.0f)" % (name, ms, *us))
x, tg, xl, tl = bench.aligned_batch(10, B, T, V, S, 10.0)
tg2, tl2 = tg.clone(), tl.clone()
for k in range(8): tg2[32 * k], tl2[32 * k] = tg[32 * k + 1], tl[32 * k + 1]
run("label_noise", (x, tg2, xl, tl2))
for scale, seed in ((3.0, 77), (8.0, 78)):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, V, generator=g) * scale
    tg = torch.randint(1, V, (B, S), generator=g); tl = torch.randint(S // 2, S + 1, (B,), generator=g)
    run("unrelated x%g" % scale, (x, tg, torch.full((B,), T, dtype=torch.long), tl))
