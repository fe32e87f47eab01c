"""What the module path adds to the C-ABI step: wall time per step with one synchronisation at the end, the CPU's
enqueue time per step (is the host ahead of the GPU?), and the kernels one step launches (torch profiler).
  python tools/diag/module_overhead.py"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
import bench
from end2end_amd import CTCLoss
d = torch.device("cuda", 0)
w = bench.WORKLOAD
_, db = bench.make_batch(1000, w["B"], w["T"], w["V"], w["S"], d)
hp = bench.HotPath(db)
crit = CTCLoss(reduce=True, size_average=True, blank_idx=0)
xm = db[0].clone().requires_grad_(); tg, xl, tl = db[1], db[2], db[3]
def module_step():
    xm.grad = None
    loss = crit(xm, tg, xl, tl); loss.backward(); return loss
def fwd_only():
    return crit(xm, tg, xl, tl)
def timed(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t2 - t0) / n * 1e6, (t1 - t0) / n * 1e6
for name, fn in (("C-ABI step (bench.HotPath.step)", hp.step), ("module forward only", fwd_only), ("module forward + backward", module_step)):
    best = min(timed(fn) for _ in range(5))
    print("%-36s %.1f us per step (host enqueue %.1f us per step)" % (name, best[0], best[1]))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5): module_step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))
