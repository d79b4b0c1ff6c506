"""Optimizer / schedule objects returned by ``configure_optimizers`` (reference clipcap/model/model.py:67-91).

``ArenaAdamW`` is a torch.optim.Optimizer whose step() is ONE fused HIP kernel per parameter arena
(torch.optim.AdamW math: betas (0.9, 0.999), eps 1e-8, weight_decay 0.01 — the defaults the reference gets from
``AdamW(self.parameters(), lr=...)``, model.py:77) and refreshes the bf16 operand copies.
``linear_warmup_decay`` is the lambda of transformers.get_linear_schedule_with_warmup (model.py:79-83).
"""
from __future__ import annotations

from typing import Callable, List

import torch


def linear_warmup_decay(num_warmup_steps: int, num_training_steps: int) -> Callable[[int], float]:
    def factor(step: int) -> float:
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))
    return factor


class ArenaAdamW(torch.optim.Optimizer):
    def __init__(self, arena_modules: List, lr: float = 2e-5, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01):
        self.arena_modules = list(arena_modules)
        params = [p for m in self.arena_modules for p in m._arena_params.values()]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._t = 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        self._t += 1
        for m in self.arena_modules:
            arena = m.engine.arena
            flat = arena.grads()
            views = m.engine.views(flat)
            for name, p in m._arena_params.items():     # gradients produced by autograd live outside the arena: gather them
                if p.grad is not None and p.grad.data_ptr() != views[name].data_ptr():
                    views[name].copy_(p.grad)
            arena.adamw_step(g["lr"], self._t, betas=g["betas"], eps=g["eps"], weight_decay=g["weight_decay"])
        return loss

    def zero_grad(self, set_to_none: bool = True):
        for m in self.arena_modules:
            if m.engine.arena.g32 is not None:
                m.engine.arena.g32.zero_()
            for p in m._arena_params.values():
                p.grad = None
