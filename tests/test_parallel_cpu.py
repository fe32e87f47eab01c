"""CPU, world_size 2, gloo: the batch-sharded reduction and gather logic (the N>1 path of bench.py / ShardedCTCLoss)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from end2end_amd.parallel import gather_decoded, reduce_sharded_losses, shard_batch, shard_bounds


def test_shard_bounds_cover_the_batch():
    for n in (0, 1, 7, 8, 4096):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, size_average, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(3)
    # a stand-in for per-utterance losses that depends on "logits" so that gradients can be checked
    logits = torch.randn(7, 5, generator=g, dtype=torch.float64)
    weights = torch.randn(7, 5, generator=g, dtype=torch.float64)
    (mine, w_mine) = shard_batch((logits, weights), rank, world)
    mine = mine.clone().requires_grad_()
    local = (mine * w_mine).sum(1) ** 2
    total = reduce_sharded_losses(local, size_average)
    total.backward()
    ref_logits = logits.clone().requires_grad_()
    ref = ((ref_logits * weights).sum(1) ** 2)
    ref_total = ref.mean() if size_average else ref.sum()
    ref_total.backward()
    lo, hi = shard_bounds(7, rank, world)
    ok = torch.allclose(total.detach(), ref_total.detach()) and torch.allclose(mine.grad, ref_logits.grad[lo:hi])
    # ragged decode gather
    ids = torch.arange((rank + 2) * (rank + 3)).reshape(rank + 2, rank + 3)
    lens = torch.full((rank + 2,), rank + 1)
    all_ids, all_lens = gather_decoded(ids, lens)
    ok = ok and [tuple(t.shape) for t in all_ids] == [(2, 3), (3, 4)] and all_lens[1].tolist() == [2, 2, 2]
    ok = ok and torch.equal(all_ids[rank], ids)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_sharded_reduction_matches_single_process():
    for size_average in (True, False):
        port = _free_port()
        with mp.Manager() as m:
            out = m.dict()
            mp.spawn(_worker, args=(2, port, size_average, out), nprocs=2, join=True)
            assert dict(out) == {0: True, 1: True}
