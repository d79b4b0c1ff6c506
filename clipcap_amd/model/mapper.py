"""TransformerMapper / TransformerMapperWindowed with the reference's constructor signature, attribute names and
state-dict keys (clipcap/model/mapper.py:113-160), executed by hand-written HIP kernels (clipcap_amd/csrc).

forward(x) -> (B, prefix_length, lm_embedding_size), identical contract to mapper.py:122-130 / :148-160.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from clipcap_amd.engine import MapperEngine
from clipcap_amd.model.arena_module import ArenaModule


class _MapperFn(torch.autograd.Function):
    """Autograd bridge for callers that use loss.backward() (the fused trainer bypasses autograd entirely)."""

    @staticmethod
    def forward(ctx, module, x, *params):
        # grad mode is always OFF inside Function.forward, so "do we need the activations" cannot be asked here: apply() is only
        # reached from TransformerMapper.forward when a backward is possible, and the forward then always keeps them (save=True).
        ctx.module = module
        ctx.batch = x.shape[0]
        eng = module.engine
        out = eng.forward(x, save=True)
        ctx.serial = eng._fwd_serial         # bumped by every saving forward of this engine (fused trainer included)
        return out

    @staticmethod
    def backward(ctx, dout):
        eng = ctx.module.engine
        if getattr(eng, "_fwd_serial", 0) != ctx.serial or getattr(eng, "_saved_batch", None) != ctx.batch:
            # the activation workspace is per engine: a later saving forward has overwritten what this graph needs
            raise RuntimeError("TransformerMapper.backward: the saved activations were overwritten by a later forward; "
                               "call backward() before running the mapper again")
        g = eng.arena.grads()
        keep = g.clone()                     # the gradient arena may hold the fused trainer's accumulation: leave it as found
        g.zero_()
        eng.backward(dout.contiguous())
        views = eng.views(g)
        outs = tuple(views[n].clone() for n in ctx.module._arena_params)
        g.copy_(keep)
        return (None, None) + outs


class TransformerMapper(ArenaModule):
    def __init__(self, encoder_embedding_size: int, lm_embedding_size: int, prefix_length: int, projection_length: int,
                 num_heads: int = 8, num_layers: int = 8, *, window_size: int = 1, use_pos_embeddings: bool = False, precision=None):
        super().__init__()
        self.projection_length = projection_length
        self.window_size = window_size
        self.engine = MapperEngine(encoder_embedding_size, lm_embedding_size, prefix_length, projection_length, num_heads, num_layers,
                                   window=window_size, use_pos=use_pos_embeddings, precision=precision)
        self._bind_parameters()
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self):
        """torch defaults of the reference modules: nn.Linear kaiming-uniform(a=sqrt 5) == U(+-1/sqrt(fan_in)) for weight and
        bias, LayerNorm (1, 0), prefix_const / pos_embeddings ~ N(0,1) (mapper.py:118-120,143)."""
        for name, p in self._arena_params.items():
            if name in ("prefix_const", "pos_embeddings"):
                p.normal_()
            elif ".norm" in name:
                p.fill_(1.0 if name.endswith("weight") else 0.0)
            elif name.endswith("weight"):
                bound = 1.0 / p.shape[1] ** 0.5
                p.uniform_(-bound, bound)
            else:
                w = self._arena_params[name[:-4] + "weight"]
                bound = 1.0 / w.shape[1] ** 0.5
                p.uniform_(-bound, bound)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._arena_params.values()):
            return _MapperFn.apply(self, x, *self._arena_params.values())
        return self.engine.forward(x, save=False)


    @torch.no_grad()
    def forward_with_attention(self, x: torch.Tensor):
        """(prefix, [attention (B, S, S, H) per layer]) — the reference's Transformer.forward_with_attention (mapper.py:45-52) seen
        from the mapper: S = window * projection_length + prefix_length rows, probabilities over the key axis (dim 2)."""
        out = self.engine.forward(x, save=True)
        return out, self.engine.attention_probs(x.shape[0])


class TransformerMapperWindowed(TransformerMapper):
    """mapper.py:133-160: input (B, window_size, E); sequence = window_size*projection_length + prefix_length."""

    def __init__(self, encoder_embedding_size: int, lm_embedding_size: int, prefix_length: int, projection_length: int, window_size: int,
                 use_pos_embeddings: bool, num_heads: int = 8, num_layers: int = 8):
        super().__init__(encoder_embedding_size, lm_embedding_size, prefix_length, projection_length, num_heads, num_layers,
                         window_size=window_size, use_pos_embeddings=use_pos_embeddings)
