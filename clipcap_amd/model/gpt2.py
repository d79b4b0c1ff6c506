"""GPT-2 language model module backed by the HIP library, API-compatible with what the reference uses of
``transformers.GPT2LMHeadModel`` (clipcap/model/model.py:19-20,45,56; inference/base.py:76,81,117):

    lm(inputs_embeds=Tensor[, attention_mask=BoolTensor]).logits     -> (B, T, V) fp32
    lm.get_input_embeddings()(ids)                                   -> (..., D)
    lm.get_input_embeddings().weight.shape[1]                        -> D
    state_dict keys == HF's (transformer.wte.weight, transformer.h.<i>.attn.c_attn.weight, ..., lm_head.weight)

A right-padding ``attention_mask`` is accepted and ignored: under the causal mask it has exactly zero effect on non-pad
rows (pinned by tests/golden/gpt2_tiny.npz 'logits_masked'), and pad rows are never read by the loss.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from clipcap_amd.engine import Gpt2Engine
from clipcap_amd.model.arena_module import ArenaModule


class _LogitsFn(torch.autograd.Function):
    """Autograd bridge for ``lm(inputs_embeds=...).logits`` (the fused trainer bypasses autograd): backward returns the gradient wrt
    inputs_embeds (which flows on into the mapper / the token embeddings) and, when the parameters require grad, every GPT-2 weight
    gradient computed by the HIP backward kernels (cc_gpt2_logits_bwd)."""

    @staticmethod
    def forward(ctx, module, x, train, *params):
        ctx.module, ctx.train = module, train
        out = module.engine.logits(x, save_mode=2 if train else 1)
        ctx.serial = module.engine._logits_pass[1]
        return out

    @staticmethod
    def backward(ctx, dlogits):
        eng = ctx.module.engine
        if eng._logits_pass[1] != ctx.serial:
            raise RuntimeError("GPT2LM.backward: the saved activations were overwritten by a later forward; call backward() first")
        if not ctx.train:
            return (None, eng.logits_backward(dlogits), None) + (None,) * len(ctx.module._arena_params)
        g = eng.arena.grads()
        keep = g.clone()
        g.zero_()
        dx0 = eng.logits_backward(dlogits)
        views = eng.views(g)
        grads = tuple(views[n].clone() for n in ctx.module._arena_params)
        g.copy_(keep)
        return (None, dx0, None) + grads


class _EmbedFn(torch.autograd.Function):
    """wte[ids] with its gradient, both on the library's kernels (cc_embed_tokens / cc_embed_tokens_bwd): the autograd path of a full
    finetune driven through Module.forward (reference model.py:44 ``get_input_embeddings()(tokens)``)."""

    @staticmethod
    def forward(ctx, weight, ids, engine):
        from clipcap_amd.engine import embed_tokens
        ids32 = ids.to(device=weight.device, dtype=torch.int32).contiguous().view(-1)
        out = torch.empty(*ids.shape, weight.shape[1], dtype=torch.float32, device=weight.device)
        embed_tokens(engine, ids32, out.view(-1, weight.shape[1]))
        ctx.save_for_backward(ids32)
        ctx.engine = engine
        return out

    @staticmethod
    def backward(ctx, dout):
        from clipcap_amd.engine import embed_tokens_bwd
        (ids32,) = ctx.saved_tensors
        dw = embed_tokens_bwd(ctx.engine, ids32, dout.contiguous().view(ids32.numel(), -1).float())
        return dw, None, None


class _Embedding(nn.Module):
    """wte lookup on the fp32 master (a gather; exact)."""

    def __init__(self, owner: "GPT2LM"):
        super().__init__()
        object.__setattr__(self, "_owner", owner)

    @property
    def weight(self) -> nn.Parameter:
        return self._owner._arena_params["transformer.wte.weight"]

    def forward(self, ids: torch.Tensor) -> torch.Tensor:
        need_grad = torch.is_grad_enabled() and self.weight.requires_grad
        w = self.weight if need_grad else self.weight.detach()
        if w.is_cuda and not need_grad and ids.numel() > 0:
            # inference / frozen-LM paths: the library's gather (cc_embed_tokens, reference inference/base.py:117,184) — exact, like F.embedding
            from clipcap_amd.engine import embed_tokens
            # the public get_input_embeddings() surface: F.embedding raises on a bad id.  Host-resident ids are checked here for free; ids
            # already on the device are NOT read back (two blocking syncs per generated token in the sampling loops) —
            # k_embed_tokens clamps them, so a bad id cannot fault, it reads the nearest valid row
            if not ids.is_cuda:
                lo, hi = int(ids.min()), int(ids.max())
                if lo < 0 or hi >= w.shape[0]:
                    raise IndexError(f"token id out of range for the {w.shape[0]}-row embedding table: min {lo}, max {hi}")
            out = torch.empty(*ids.shape, w.shape[1], dtype=torch.float32, device=w.device)
            embed_tokens(self._owner.engine, ids.to(device=w.device, dtype=torch.int32).contiguous().view(-1), out.view(-1, w.shape[1]))
            return out
        if w.is_cuda and ids.numel() > 0:
            # autograd path (a full finetune through Module.forward): the same gather, and its scatter-add gradient, on the library's kernels
            if not ids.is_cuda:
                lo, hi = int(ids.min()), int(ids.max())
                if lo < 0 or hi >= w.shape[0]:
                    raise IndexError(f"token id out of range for the {w.shape[0]}-row embedding table: min {lo}, max {hi}")
            return _EmbedFn.apply(w, ids, self._owner.engine)
        return torch.nn.functional.embedding(ids.to(w.device), w)       # CPU master (no device): torch's gather, exact


class _TiedHead(nn.Module):
    """lm_head whose weight is the wte parameter itself (HF ties them: modeling_gpt2.py:638)."""

    def __init__(self, wte: nn.Parameter):
        super().__init__()
        self.weight = wte


class GPT2LM(ArenaModule):
    def __init__(self, n_embd: int = 768, n_layer: int = 12, n_head: int = 12, vocab_size: int = 50257, n_positions: int = 1024,
                 initializer_range: float = 0.02, name_or_path: str = "", embd_pdrop: float = 0.1, attn_pdrop: float = 0.1,
                 resid_pdrop: float = 0.1, precision=None):
        super().__init__()
        # *_pdrop: GPT2Config defaults (0.1); they act only in train mode of a full finetune (ClipCapModel), through the HIP kernels'
        # counter-based masks (cc_gpt2_shape.p_embd / p_attn / p_resid / drop_seed)
        self.config = SimpleNamespace(n_embd=n_embd, n_layer=n_layer, n_head=n_head, vocab_size=vocab_size, n_positions=n_positions,
                                      initializer_range=initializer_range, name_or_path=name_or_path, embd_pdrop=embd_pdrop,
                                      attn_pdrop=attn_pdrop, resid_pdrop=resid_pdrop)
        self.engine = Gpt2Engine(n_embd, n_head, n_layer, vocab_size, n_positions, precision=precision)
        self._bind_parameters()
        self.lm_head = _TiedHead(self._arena_params["transformer.wte.weight"])
        self._emb = _Embedding(self)
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self):
        """GPT2PreTrainedModel._init_weights: N(0, initializer_range) weights/embeddings, zero biases, LayerNorm (1,0) and the
        residual projections scaled by 1/sqrt(2*n_layer)."""
        std = self.config.initializer_range
        for name, p in self._arena_params.items():
            if ".ln_" in name:
                p.fill_(1.0 if name.endswith("weight") else 0.0)
            elif name.endswith("bias"):
                p.zero_()
            elif name.endswith("c_proj.weight"):
                p.normal_(0.0, std / (2 * self.config.n_layer) ** 0.5)
            else:
                p.normal_(0.0, std)

    # ---- construction from disk / hub (reference: AutoModelForCausalLM.from_pretrained, model.py:19) ----
    @classmethod
    def from_config_dict(cls, cfg: dict, name: str = "") -> "GPT2LM":
        return cls(n_embd=cfg.get("n_embd", 768), n_layer=cfg.get("n_layer", 12), n_head=cfg.get("n_head", 12),
                   vocab_size=cfg.get("vocab_size", 50257), n_positions=cfg.get("n_positions", 1024),
                   initializer_range=cfg.get("initializer_range", 0.02), name_or_path=name, embd_pdrop=cfg.get("embd_pdrop", 0.1),
                   attn_pdrop=cfg.get("attn_pdrop", 0.1), resid_pdrop=cfg.get("resid_pdrop", 0.1))

    @classmethod
    def from_pretrained(cls, name_or_path: str) -> "GPT2LM":
        if os.path.isdir(name_or_path):
            with open(os.path.join(name_or_path, "config.json")) as f:
                cfg = json.load(f)
            if cfg.get("model_type", "gpt2") != "gpt2":
                raise ValueError(f"only GPT-2 family language models are implemented in HIP (got {cfg.get('model_type')})")
            model = cls.from_config_dict(cfg, name_or_path)
            sd = None
            st = os.path.join(name_or_path, "model.safetensors")
            pt = os.path.join(name_or_path, "pytorch_model.bin")
            if os.path.exists(st):
                from safetensors.torch import load_file
                sd = load_file(st)
            elif os.path.exists(pt):
                sd = torch.load(pt, map_location="cpu")
            if sd is not None:
                sd = {(k if k.startswith(("transformer.", "lm_head.")) else "transformer." + k): v for k, v in sd.items()}
                model.load_state_dict(sd, strict=False)
            return model
        # hub name: let transformers resolve/download it (needs network or a populated HF cache), then copy the weights
        from transformers import AutoModelForCausalLM
        hf = AutoModelForCausalLM.from_pretrained(name_or_path)
        c = hf.config
        model = cls(n_embd=c.n_embd, n_layer=c.n_layer, n_head=c.n_head, vocab_size=c.vocab_size, n_positions=c.n_positions,
                    initializer_range=c.initializer_range, name_or_path=name_or_path, embd_pdrop=c.embd_pdrop, attn_pdrop=c.attn_pdrop,
                    resid_pdrop=c.resid_pdrop)
        model.load_state_dict(hf.state_dict(), strict=False)
        return model

    def save_pretrained(self, path: str):
        os.makedirs(path, exist_ok=True)
        c = self.config
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(dict(model_type="gpt2", architectures=["GPT2LMHeadModel"], n_embd=c.n_embd, n_layer=c.n_layer, n_head=c.n_head,
                           vocab_size=c.vocab_size, n_positions=c.n_positions, n_ctx=c.n_positions, initializer_range=c.initializer_range,
                           activation_function="gelu_new", layer_norm_epsilon=1e-5), f)
        from safetensors.torch import save_file
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items() if k != "lm_head.weight"},
                  os.path.join(path, "model.safetensors"))

    # ---- the calls the reference makes ----
    def get_input_embeddings(self) -> _Embedding:
        return self._emb

    def forward(self, inputs_embeds: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
                input_ids: Optional[torch.Tensor] = None, **unused):
        if inputs_embeds is None:
            if input_ids is None:
                raise ValueError("GPT2LM.forward needs inputs_embeds or input_ids")
            inputs_embeds = self._emb(input_ids)
        if torch.is_grad_enabled():
            train = any(p.requires_grad for p in self._arena_params.values())
            if train or inputs_embeds.requires_grad:
                return SimpleNamespace(logits=_LogitsFn.apply(self, inputs_embeds, train, *self._arena_params.values()))
        return SimpleNamespace(logits=self.engine.logits(inputs_embeds))
