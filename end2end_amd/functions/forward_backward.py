"""Autograd glue for forward-backward style losses.

Same surface as the reference's ForwardBackwardLossFunction
(pytorch_end2end/functions/forward_backward.py:4-35): forward asks the engine for
(loss, grads) in one pass and keeps the gradient; backward scales it by grad_output.
Two optional trailing arguments extend it:
  fused_logits  the engine fuses log-softmax into the kernel; the kept gradient is d loss / d logits;
  reduction     "sum" / "mean": the Function returns the reduced loss itself (what the module's
                `loss.sum()` / `loss.mean()` gives, pytorch_end2end/modules/ctc_loss.py:52-56), computed by
                the tail of the engine's last kernel, and the kept gradient is already scaled by 1/B for
                the mean -- the backward of the usual `loss.backward()` then has nothing left to multiply.

The backward does not build a second (B,T,V) tensor: the kept gradient is scaled in place by the
library (e2e_ctc_scale_grads; rows whose factor is exactly 1 are not touched) and handed to autograd,
which can then take it as `.grad` without a copy.  Upstream the kept gradient survives the backward, so
a graph retained with `retain_graph=True` can be walked again; here a second walk finds the buffer given
away and asks the engine for it again (same inputs, same result) -- rare, and exact.
"""
import torch
from torch.autograd import Function


class ForwardBackwardLossFunction(Function):
    @staticmethod
    def _compute(engine, args, fused_logits, reduction):
        logits, targets, logits_lengths, targets_lengths = args
        if reduction is None and not fused_logits:         # exactly the reference's call
            return engine.compute(logits, targets, logits_lengths, targets_lengths) + (None,)
        kw = {"input_is_logprobs": not fused_logits}
        if reduction is not None:
            kw["reduction"] = reduction
            kw["grad_scale"] = 1.0 / max(logits.shape[0], 1) if reduction == "mean" else 1.0
            return engine.compute(logits, targets, logits_lengths, targets_lengths, **kw)
        return engine.compute(logits, targets, logits_lengths, targets_lengths, **kw) + (None,)

    @staticmethod
    def forward(ctx, engine, logits, targets, logits_lengths, targets_lengths, fused_logits=False, reduction=None):
        args = (logits.detach(), targets, logits_lengths, targets_lengths)
        loss, grads, reduced = ForwardBackwardLossFunction._compute(engine, args, fused_logits, reduction)
        ctx.engine, ctx.args = engine, args
        ctx.fused_logits, ctx.reduction = fused_logits, reduction
        ctx.grads = grads          # plain attribute, as in the reference (no double backward)
        return loss if reduction is None else reduced

    @staticmethod
    def backward(ctx, grad_output):
        grads = ctx.grads
        if grads is None:          # a retained graph walked again: the first walk gave the buffer to autograd
            grads = ForwardBackwardLossFunction._compute(ctx.engine, ctx.args, ctx.fused_logits, ctx.reduction)[1]
        ctx.grads = None
        if (grads.is_cuda and grads.is_contiguous()
                and grads.dtype in (torch.float32, torch.float64, torch.float16, torch.bfloat16)
                and hasattr(ctx.engine, "scale_grads_")):
            ctx.engine.scale_grads_(grads, grad_output)      # (16-bit gradients: multiplied in their own dtype, as upstream)
        else:                      # results moved back to a CPU source tensor
            go = grad_output.contiguous().to(device=grads.device, dtype=grads.dtype)
            grads = grads * (go.view(-1, 1, 1) if go.numel() > 1 else go)
        if grads.device != grad_output.device:
            grads = grads.to(grad_output.device)
        return None, grads, None, None, None, None, None
