"""Autograd glue for forward-backward style losses.

Same surface as the reference's ForwardBackwardLossFunction
(pytorch_end2end/functions/forward_backward.py:4-35): forward asks the engine for
(loss, grads) in one pass and keeps the gradient; backward scales it by grad_output.
The extra `fused_logits` flag selects the engine mode in which log-softmax is fused into
the kernel and the stored gradient is already d loss / d logits.

The backward does not build a second (B,T,V) tensor: the kept gradient is scaled in place by the
library (e2e_ctc_scale_grads; rows whose factor is exactly 1 are not touched) and handed to autograd,
which can then take it as `.grad` without a copy.  Upstream the kept gradient survives the backward, so
a graph retained with `retain_graph=True` can be walked again; here a second walk finds the buffer given
away and asks the engine for it again (same inputs, same result) -- rare, and exact.
"""
from torch.autograd import Function


class ForwardBackwardLossFunction(Function):
    @staticmethod
    def forward(ctx, engine, logits, targets, logits_lengths, targets_lengths, fused_logits=False):
        if fused_logits:
            loss, grads = engine.compute(logits, targets, logits_lengths, targets_lengths, input_is_logprobs=False)
        else:
            loss, grads = engine.compute(logits, targets, logits_lengths, targets_lengths)
        ctx.engine = engine
        ctx.fused_logits = fused_logits
        ctx.grads = grads          # plain attribute, as in the reference (no double backward)
        ctx.args = (logits.detach(), targets, logits_lengths, targets_lengths)
        return loss

    @staticmethod
    def backward(ctx, grad_output):
        grads = ctx.grads
        if grads is None:          # a retained graph walked again: the first walk gave the buffer to autograd
            x, tg, xl, tl = ctx.args
            grads = (ctx.engine.compute(x, tg, xl, tl, input_is_logprobs=False) if ctx.fused_logits
                     else ctx.engine.compute(x, tg, xl, tl))[1]
        ctx.grads = None
        if grads.is_cuda and grads.is_contiguous() and hasattr(ctx.engine, "scale_grads_"):
            ctx.engine.scale_grads_(grads, grad_output)
        else:                      # results that were moved back to a CPU source tensor
            grads = grads * grad_output.contiguous().to(device=grads.device, dtype=grads.dtype).view(-1, 1, 1)
        if grads.device != grad_output.device:
            grads = grads.to(grad_output.device)
        return None, grads, None, None, None, None
