"""Optimiser step of the training config on libmvq_hip.so: ``clip_grad_norm_`` and ``AdamW`` with the call shapes of
``torch.nn.utils.clip_grad_norm_`` / ``torch.optim.AdamW`` (Training/compare_dacvsproposal_5.py:367,394-395).

The reference's own calls work on these modules unchanged (the parameters are ordinary ``nn.Parameter``s); this module is the
MI355X-native alternative: one sum-of-squares launch per tensor with a single device-side combine (no host sync), and one
fused update launch per tensor that applies the clip factor on the fly instead of rewriting the gradients.
"""
from __future__ import annotations

import torch

from . import _lib, ops


def grad_norm(params):
    """Global L2 norm of the gradients as a 0-d device tensor (no host sync)."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return None
    dev = grads[0].device
    P = 64
    partial = torch.zeros(len(grads), P, device=dev, dtype=torch.float32)
    for i, g in enumerate(grads):
        g = ops._dev(g.contiguous(), "grad")
        ops.check(_lib.lib().mvq_sumsq_partial_f32(g.data_ptr(), partial.data_ptr() + 4 * i * P, P, g.numel(), ops._stream()),
                  "mvq_sumsq_partial_f32")
    return ops.rowsum(partial.reshape(1, -1)).reshape(()).sqrt()


def clip_coef(params, max_norm: float):
    """(total_norm, coef) with coef = min(1, max_norm / (total_norm + 1e-6)) -- torch.nn.utils.clip_grad_norm_'s factor."""
    total = grad_norm(params)
    if total is None:
        return None, None
    return total, (float(max_norm) / (total + 1e-6)).clamp(max=1.0).reshape(1).contiguous()


@torch.no_grad()
def clip_grad_norm_(params, max_norm: float):
    """Drop-in for torch.nn.utils.clip_grad_norm_ (L2): scales the gradients in place, returns the total norm (device)."""
    params = [p for p in params if p.grad is not None]
    total, coef = clip_coef(params, max_norm)
    if total is None:
        return torch.zeros(())
    for p in params:
        p.grad.mul_(coef[0])
    return total


class AdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction, eps outside the sqrt), one fused HIP launch per
    tensor.  ``step(clip_coef=...)`` fuses the gradient clip: pass the tensor from ``clip_coef(params, max_norm)``."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None, clip_coef=None):
        loss = closure() if closure is not None else None
        f = _lib.lib().mvq_adamw_f32
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = ops._dev(p.grad.contiguous(), "grad")
                ops._dev(p.data, "param")
                ops.check(f(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                            ops._p(clip_coef), p.numel(), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                            float(group["weight_decay"]), int(st["step"]), ops._stream()), "mvq_adamw_f32")
                torch.autograd.graph.increment_version(p)        # written through the raw pointer: packed-weight caches must refresh
        return loss
