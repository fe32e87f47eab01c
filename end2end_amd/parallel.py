"""Batch-sharded CTC across the GPUs of a node: one process per GPU (torch.distributed, backend "nccl" = RCCL).

Utterances are independent (src/losses/forward_backward.cpp:38-52, src/decoders/ctc_decoder.cpp:174-189), so every rank
runs the kernels on its own slice of the batch and nothing crosses the fabric on the data path.  The only exchange is
the scalar loss: one all-reduce of [sum of losses, number of utterances] (16 bytes, latency-bound over xGMI).
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world_size):
    """Contiguous, balanced slice [lo, hi) of n_items for `rank` (the first n_items % world_size ranks get one more)."""
    q, r = divmod(n_items, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_batch(tensors, rank, world_size, dim=0):
    """Slice every tensor of a batch along the utterance dimension for this rank."""
    n = tensors[0].shape[dim]
    lo, hi = shard_bounds(n, rank, world_size)
    return tuple(t.narrow(dim, lo, hi - lo) for t in tensors)


def reduce_sharded_losses(local_losses, size_average=True, group=None):
    """Global mean (or sum) of per-utterance losses that are sharded over ranks.

    The value is the global reduction on every rank; the gradient flows only into the local losses, scaled as the
    global reduction scales them (1/N_global for the mean), so each rank's logits.grad equals what a single-process
    run over the whole batch would give for its utterances.
    """
    local_sum = local_losses.sum()
    stats = torch.stack([local_sum.detach().to(torch.float64),
                         torch.tensor(float(local_losses.numel()), dtype=torch.float64, device=local_losses.device)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    denom = stats[1] if size_average else torch.ones_like(stats[1])
    value = (stats[0] / denom).to(local_losses.dtype)
    return value.detach() + (local_sum - local_sum.detach()) / denom.to(local_losses.dtype)


class ShardedCTCLoss(torch.nn.Module):
    """CTCLoss over a batch whose utterances are spread over the ranks of `group`; returns the GLOBAL mean/sum."""

    def __init__(self, size_average=True, after_logsoftmax=False, time_major=False, blank_idx=0, group=None):
        super().__init__()
        from .modules.ctc_loss import CTCLoss
        self._ctc = CTCLoss(size_average=None, reduce=None, after_logsoftmax=after_logsoftmax,
                            time_major=time_major, blank_idx=blank_idx)
        self._size_average = size_average
        self._group = group

    def forward(self, logits, targets, logits_lengths, targets_lengths):
        local = self._ctc(logits, targets, logits_lengths, targets_lengths)
        return reduce_sharded_losses(local, self._size_average, self._group)


def gather_decoded(decoded_targets, decoded_lengths, group=None):
    """all_gather of ragged decode results (lengths + right-padded label matrix) -> lists ordered by rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [decoded_targets], [decoded_lengths]
    world = dist.get_world_size(group)
    shape = torch.tensor(list(decoded_targets.shape), dtype=torch.long, device=decoded_targets.device)
    shapes = [torch.zeros_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape, group=group)
    rows = max(int(s[0]) for s in shapes)
    width = max(int(s[1]) for s in shapes)
    pad = torch.zeros((rows, width), dtype=decoded_targets.dtype, device=decoded_targets.device)
    pad[: decoded_targets.shape[0], : decoded_targets.shape[1]] = decoded_targets
    lens = torch.zeros(rows, dtype=decoded_lengths.dtype, device=decoded_lengths.device)
    lens[: decoded_lengths.shape[0]] = decoded_lengths
    all_t = [torch.zeros_like(pad) for _ in range(world)]
    all_l = [torch.zeros_like(lens) for _ in range(world)]
    dist.all_gather(all_t, pad, group=group)
    dist.all_gather(all_l, lens, group=group)
    return ([t[: int(s[0]), : int(s[1])] for t, s in zip(all_t, shapes)],
            [l[: int(s[0])] for l, s in zip(all_l, shapes)])
