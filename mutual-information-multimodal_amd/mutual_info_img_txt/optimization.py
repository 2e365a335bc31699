"""The two optimisation pieces the reference trainer takes from a third-party package that is absent here:
``pytorch_transformers.optimization.AdamW`` and ``WarmupLinearSchedule`` (pytorch-transformers==1.0.0, pinned in the
reference's conda_environment.yml:40; used at main_utils.py:14,166-172).  Restated from that package's published
algorithm; the arithmetic is pinned by tests/test_trainer.py against the numpy restatement in oracle/mi_oracle.py.

AdamW as published there (``correct_bias=False`` is what the reference passes, as the original BERT code does):

    m <- b1 m + (1 - b1) g ;  v <- b2 v + (1 - b2) g*g
    step = lr                       (correct_bias=False)
         = lr sqrt(1 - b2^t) / (1 - b1^t)   (correct_bias=True)
    p <- p - step * m / (sqrt(v) + eps)
    p <- p - lr * weight_decay * p          (decoupled decay, AFTER the Adam update, on the updated p)

Defaults: betas (0.9, 0.999), eps 1e-6 (not torch's 1e-8), weight_decay 0.  This differs from ``torch.optim.AdamW``
(always bias-corrected, decay applied before the update), so it is written out here.  Host-side bookkeeping on a few
hundred small tensors: plain torch ops on the parameters' device, nothing for a HIP kernel to win.

Upstream notice (pytorch-transformers 1.0.0, ``optimization.py``): Copyright 2018 The Google AI Language Team Authors and
The HuggingFace Inc. team.  Licensed under the Apache License, Version 2.0 (http://www.apache.org/licenses/LICENSE-2.0);
the interface (class names, argument names, defaults) and the update rule restated here follow that file."""
from __future__ import annotations

import math

import torch
from torch.optim import Optimizer
from torch.optim.lr_scheduler import LambdaLR


class AdamW(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        if lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr} - should be >= 0.0")
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameters: {betas} - should be in [0.0, 1.0[")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps} - should be >= 0.0")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                state = self.state[p]
                if not state:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p)
                    state["exp_avg_sq"] = torch.zeros_like(p)
                state["step"] += 1
                m, v = state["exp_avg"], state["exp_avg_sq"]
                m.mul_(b1).add_(p.grad, alpha=1.0 - b1)
                v.mul_(b2).addcmul_(p.grad, p.grad, value=1.0 - b2)
                step_size = group["lr"]
                if group["correct_bias"]:
                    t = state["step"]
                    step_size = step_size * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
                p.addcdiv_(m, v.sqrt().add_(group["eps"]), value=-step_size)
                if group["weight_decay"] > 0.0:
                    p.add_(p, alpha=-group["lr"] * group["weight_decay"])
        return loss


def warmup_linear_factor(step: int, warmup_steps: float, t_total: float) -> float:
    """Learning-rate multiplier of WarmupLinearSchedule: linear 0 -> 1 over warmup_steps, then linear 1 -> 0 at t_total."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(t_total - step) / float(max(1.0, t_total - warmup_steps)))


class WarmupLinearSchedule(LambdaLR):
    def __init__(self, optimizer, warmup_steps, t_total, last_epoch=-1):
        self.warmup_steps = warmup_steps
        self.t_total = t_total
        super().__init__(optimizer, lambda step: warmup_linear_factor(step, warmup_steps, t_total), last_epoch=last_epoch)
