"""Autograd glue for forward-backward style losses.

Same surface as the reference's ForwardBackwardLossFunction
(pytorch_end2end/functions/forward_backward.py:4-35): forward asks the engine for
(loss, grads) in one pass and keeps the gradient; backward scales it by grad_output.
The extra `fused_logits` flag selects the engine mode in which log-softmax is fused into
the kernel and the stored gradient is already d loss / d logits.
"""
import torch
from torch.autograd import Function


class ForwardBackwardLossFunction(Function):
    @staticmethod
    def forward(ctx, engine, logits, targets, logits_lengths, targets_lengths, fused_logits=False):
        if fused_logits:
            loss, grads = engine.compute(logits, targets, logits_lengths, targets_lengths, input_is_logprobs=False)
        else:
            loss, grads = engine.compute(logits, targets, logits_lengths, targets_lengths)
        ctx.grads = grads          # plain attribute, as in the reference (no double backward)
        return loss

    @staticmethod
    def backward(ctx, grad_output):
        grads = ctx.grads
        if grads.device != grad_output.device:
            grads = grads.to(grad_output.device)
        scale = grad_output.contiguous().to(grads.dtype).view(-1, 1, 1)
        return None, grads * scale, None, None, None, None
