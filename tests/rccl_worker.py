"""Worker of tests/test_gpu_rccl.py: started by torch.distributed.run, one process per GPU, backend "nccl" (= RCCL).
ShardedCTCLoss on the real kernels against the unsharded module on the same GPU; prints RCCL_OK from rank 0."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from end2end_amd import CTCLoss                                            # noqa: E402
from end2end_amd.parallel import ShardedCTCLoss, gather_decoded, shard_batch, shard_bounds   # noqa: E402


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    # the collective itself, also at world size 1 (where the sharded loss skips it)
    t = torch.tensor([rank + 1.0, 1.0], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    assert t.tolist() == [world * (world + 1) / 2.0, float(world)], t.tolist()

    g = torch.Generator().manual_seed(11)
    B, T, V, S = 7, 60, 13, 9
    x = torch.randn(B, T, V, generator=g)
    tg = torch.randint(1, V, (B, S), generator=g)
    xl = torch.tensor([60, 51, 60, 44, 60, 37, 60])
    tl = torch.tensor([9, 4, 0, 7, 9, 5, 8])
    for size_average in (True, False):
        full = x.clone().to(dev).requires_grad_()
        ref = CTCLoss(reduce=True, size_average=size_average)(full, tg.to(dev), xl.to(dev), tl.to(dev))
        ref.backward()
        mine = tuple(t_.contiguous().to(dev) for t_ in shard_batch((x, tg, xl, tl), rank, world))
        xs = mine[0].clone().requires_grad_()
        tot = ShardedCTCLoss(size_average=size_average)(xs, mine[1], mine[2], mine[3])
        tot.backward()
        lo, hi = shard_bounds(B, rank, world)
        assert abs(tot.item() - ref.item()) <= 1e-6 * abs(ref.item()), (tot.item(), ref.item())
        assert torch.allclose(xs.grad, full.grad[lo:hi], rtol=1e-6, atol=1e-9), (xs.grad - full.grad[lo:hi]).abs().max().item()
    ids = torch.arange((rank + 2) * 3, device=dev).reshape(rank + 2, 3)
    all_ids, all_lens = gather_decoded(ids, torch.full((rank + 2,), 3, device=dev))
    assert len(all_ids) == world and torch.equal(all_ids[rank], ids)
    dist.barrier()
    if rank == 0:
        print("RCCL_OK world=%d" % world, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
