"""Does replaying the loss call (+ batch mean) from a captured hipGraph beat launching its kernels one by one?"""
import sys, os, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch, bench
dev = torch.device("cuda", 0)
w = bench.WORKLOAD
host, devb = bench.make_batch(1000, w["B"], w["T"], w["V"], w["S"], dev)
hp = bench.HotPath(devb)
for _ in range(5): hp.step()
torch.cuda.synchronize()
def timed(fn, n=200):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, e0.elapsed_time(e1) / n * 1e3
print("plain launches: wall %.1f us, events %.1f us per step" % timed(hp.step))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    hp.k = 0; hp.step(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    hp.k = 0
    with torch.cuda.graph(g, stream=s):
        hp.step()
torch.cuda.synchronize()
print("graph replay:   wall %.1f us, events %.1f us per step" % timed(g.replay))
