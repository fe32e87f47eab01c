"""CTCLoss module with the reference's constructor and call signature
(pytorch_end2end/modules/ctc_loss.py:15-75), computing on the MI355X.

Differences that do not change results:
  * with after_logsoftmax=False the log-softmax (and its backward) is fused into the HIP
    kernel instead of running F.log_softmax before the engine (`fused=False` restores the
    reference's two-step structure);
  * nothing is copied to the host.
"""
import torch.nn as nn
import torch.nn.functional as F

from ..engines import CTCLossEngine
from ..functions.forward_backward import ForwardBackwardLossFunction


class ForwardBackwardLossBase(nn.Module):
    def __init__(self, size_average=None, reduce=None, after_logsoftmax=False, time_major=False, blank_idx=0):
        super().__init__()
        self._blank_idx = blank_idx
        self._reduce = reduce
        self._size_average = size_average
        self._after_logsoftmax = after_logsoftmax
        self._time_major = time_major
        self._engine = None
        self._fused = True

    def forward(self, logits, targets, logits_lengths, targets_lengths):
        """
        :param logits: float tensor, ``(time, batch, alphabet)`` if ``time_major`` else ``(batch, time, alphabet)``
        :param targets: ``(batch, max_target_length)`` integer tensor
        :param logits_lengths: ``(batch,)`` frame counts
        :param targets_lengths: ``(batch,)`` target lengths
        :return: ``(batch,)`` losses when ``reduce`` is falsy, else their mean (``size_average``) or sum
        """
        fuse = self._fused and not self._after_logsoftmax
        x = logits
        if not self._after_logsoftmax and not fuse:
            x = F.log_softmax(x, dim=2)
        if self._time_major:
            x = x.permute(1, 0, 2)          # a strided view; the kernel reads through the strides
        # reduce: the sum / mean is written by the tail of the engine's last kernel (no separate reduction) and the
        # kept gradient is already scaled, so `loss.backward()` finds nothing left to multiply
        reduction = ("mean" if self._size_average else "sum") if self._reduce else None
        return ForwardBackwardLossFunction.apply(self._engine, x, targets, logits_lengths, targets_lengths, fuse,
                                                 reduction)


class CTCLoss(ForwardBackwardLossBase):
    """
    Connectionist Temporal Classification loss (Graves et al., 2006).

    :param size_average: average (instead of sum) over the batch; only with ``reduce``
    :param reduce: reduce to a scalar; ``None`` returns the ``(batch,)`` vector
    :param after_logsoftmax: inputs are already log-probabilities
    :param time_major: inputs are ``(time, batch, alphabet)``
    :param blank_idx: index of the blank label
    :param fused: fuse log-softmax into the kernel when ``after_logsoftmax`` is False
    :param f32_chains: (extension, default off) let the lattice chains of long-target batches run in packed f32:
        ~10 % faster at B=256, T=1000, V=29, S<=200; gradient elements within 2e-5 absolute of the reference's
        instead of 2e-6 (``e2e_ctc_loss_opts.chains`` in include/e2e_ctc.h)
    """

    def __init__(self, size_average=None, reduce=None, after_logsoftmax=False, time_major=False, blank_idx=0,
                 fused=True, f32_chains=False):
        super().__init__(size_average, reduce, after_logsoftmax, time_major, blank_idx)
        self._fused = fused
        self._engine = CTCLossEngine(self._blank_idx, f32_chains=f32_chains)


class GramCTCLoss(CTCLoss):
    """Gram-CTC is an unfinished stub upstream (empty compute_2d, src/losses/gram_ctc_loss.cpp:30-37;
    the module raises, pytorch_end2end/modules/ctc_loss.py:94-106) and is out of scope here."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("Gram-CTC is not implemented (neither is it upstream)")
