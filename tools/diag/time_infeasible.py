"""The headline batch with a few utterances that have fewer frames than labels (no alignment exists): cost of a call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
d = torch.device("cuda", 0)
for nbad in (0, 1, 8):
    _, db = bench.make_batch(1000, 256, 1000, 29, 200, d)
    x, tg, xl, tl = db
    xl = xl.clone(); xl[:nbad] = 90                     # target lengths are >= 100
    hp = bench.HotPath((x, tg, xl, tl))
    for _ in range(3): hp.call()
    ms = bench.time_events(torch, hp.call, 10)
    print("%d utterances without an alignment: %.3f ms per call, losses inf: %d" % (nbad, ms, int(torch.isinf(hp.losses).sum())))
