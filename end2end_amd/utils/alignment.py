"""Viterbi forced alignment with the reference's call signature (pytorch_end2end/utils/alignment.py:109-138),
computed on the MI355X by e2e_ctc_align instead of numba-jitted Python loops on the host.

    targets_aligned = get_alignment_3d(log_probs, targets, logits_lengths, targets_lengths, is_ctc=True)

`log_probs` is (batch, time, alphabet) AFTER log-softmax; the result is a CPU int64 tensor (batch, time) holding, for
every frame of every utterance, the label (or the blank, id 0 as upstream) it is aligned to, and -100 past the
utterance's length -- exactly what upstream returns.  `keep_on_device=True` (an extension) leaves it on the GPU.
"""
import torch

from .. import _runtime as R
from .._runtime import _C


def get_alignment_3d(log_probs, targets, logits_lengths, targets_lengths, is_ctc=True, blank_idx=0, keep_on_device=False):
    if log_probs.dim() != 3:
        raise ValueError("log_probs must be (batch, time, alphabet)")
    dev = R.compute_device(log_probs)
    x = log_probs.detach()
    if x.dtype not in (torch.float32, torch.float64, torch.float16, torch.bfloat16):      # (16-bit log-probabilities are read as they are)
        x = x.to(torch.float32)
    x = x.to(dev)
    B, T, V = x.shape
    tg = torch.as_tensor(targets).to(device=dev, dtype=torch.long)
    if tg.dim() != 2 or tg.shape[0] != B:
        raise ValueError("targets must be (batch, max_target_length)")
    if tg.shape[1] == 0:
        tg = torch.zeros((B, 1), dtype=torch.long, device=dev)
    tg = tg.contiguous()
    xl = torch.as_tensor(logits_lengths).to(device=dev, dtype=torch.long).contiguous()
    tl = torch.as_tensor(targets_lengths).to(device=dev, dtype=torch.long).contiguous()
    if xl.numel() != B or tl.numel() != B:
        raise ValueError("lengths must have one entry per utterance")
    out = torch.empty((B, T), dtype=torch.long, device=dev)
    if B:
        with torch.cuda.device(dev):
            Smax = tg.shape[1]
            ws = R.workspace(dev, _C.ctc_align_workspace_bytes(B, T, V, Smax, bool(is_ctc)))
            sB, sT, sV = x.stride()
            _C.ctc_align(x.data_ptr(), R.dtype_code(x.dtype), sB, sT, sV, tg.data_ptr(), tg.stride(0), xl.data_ptr(),
                         tl.data_ptr(), B, T, V, Smax, int(blank_idx), bool(is_ctc), out.data_ptr(), -100,
                         ws.data_ptr(), ws.numel(), R.stream_handle(dev))
    return out if keep_on_device else out.cpu()
