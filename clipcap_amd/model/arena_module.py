"""nn.Module plumbing shared by the mapper and GPT-2 modules: parameters are views of an engine's flat fp32 arena.

Why: the HIP library works on flat arenas (one fused AdamW launch, one all-reduce, bf16 operand copies refreshed in one
call) while the reference's callers expect ordinary nn.Modules whose ``state_dict()`` carries the reference's key names
(clipcap/model/load.py:34 ``load_state_dict(strict=False)``).  Both hold: each nn.Parameter's storage IS a slice of the
arena, ``.to(device)`` moves the arena and re-points the parameters, and in-place updates of a parameter (load_state_dict, torch optimizers, copy_) are
noticed through the version counters the arena watches, so the bf16 operand copy is refreshed lazily.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn as nn


class ArenaModule(nn.Module):
    """Owns ``self.engine`` (MapperEngine / Gpt2Engine) and a tree of container modules holding the parameter views."""

    def _bind_parameters(self):
        self._arena_params: Dict[str, nn.Parameter] = {}
        views = self.engine.views(self.engine.arena.w32)
        for name, view in views.items():
            *path, leaf = name.split(".")
            mod = self
            for part in path:
                if not hasattr(mod, part):
                    mod.add_module(part, nn.Module())
                mod = getattr(mod, part)
            p = nn.Parameter(view, requires_grad=True)
            mod.register_parameter(leaf, p)
            self._arena_params[name] = p
        self.engine.arena.watch = list(self._arena_params.values())

    def _rebind(self):
        views = self.engine.views(self.engine.arena.w32)
        for name, p in self._arena_params.items():
            p.data = views[name]
            p.grad = None
        # p.data = view gives every parameter its own version counter: the arena sums them to notice in-place writes
        self.engine.arena.watch = list(self._arena_params.values())

    def _apply(self, fn, recurse=True):
        # learn the target device from fn (module.to / .cuda / .cpu); dtype changes are ignored: the master stays fp32
        probe = fn(torch.empty(0, dtype=torch.float32, device=self.engine.arena.device))
        if probe.device != self.engine.arena.device:
            self.engine.to(probe.device)
            self._rebind()
        return self

    def bind_grads(self):
        """Point every parameter's .grad at its slice of the engine's gradient arena (fused training path)."""
        g = self.engine.views(self.engine.arena.grads())
        for name, p in self._arena_params.items():
            p.grad = g[name]

    def set_precision(self, precision):
        """16 = fp16 GEMM / attention operands (the reference's ``--fp-precision 16``, clipcap/train/args.py:30-34; training then
        runs under a dynamic loss scale); 32 (the reference's default) / 64 = split-bf16 operands (three MFMA terms per product, fp32
        activations: logits within 1e-3 of the fp32 reference); "bf16" = bf16 operands.  Master weights, accumulation and the
        optimizer stay fp32 (_lib.op_dtype_of)."""
        self.engine.set_precision(precision)
        return self

    @property
    def device(self):
        return self.engine.arena.device
