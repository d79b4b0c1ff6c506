"""ClipCapModel / ClipCapModelPrefixOnly with the reference's surface (clipcap/model/model.py:13-123):

    .transformer_mapper, .language_model, .config, .lm_embedding_size
    forward(tokens, embeddings, mask) -> obj with .logits            (model.py:43-58)
    training_step((tokens, embeds), idx) -> loss                     (model.py:94-113)
    configure_optimizers() -> {"optimizer", "lr_scheduler": {...}}    (model.py:67-91)
    set_training_config(cfg, reinit_optims=False)                    (model.py:60-65)
    ClipCapModelPrefixOnly.parameters() -> mapper only; .train() keeps the LM in eval (model.py:116-123)

No pytorch_lightning dependency: a plain nn.Module plus ``fused_step`` (forward+backward+AdamW as straight kernel chains),
which is what clipcap_amd.train drives.  ``training_step`` still returns a loss that supports ``.backward()`` for callers
written against the reference.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional, Tuple

import torch
import torch.nn as nn

from clipcap_amd.engine import ClipCapEngine
from clipcap_amd.model.config import Config, TrainingConfig
from clipcap_amd.model.gpt2 import GPT2LM
from clipcap_amd.model.mapper import TransformerMapper, TransformerMapperWindowed
from clipcap_amd.model.optim import ArenaAdamW, linear_warmup_decay


def get_tokenizer(language_model_name: str, **huggingface_kwargs):
    """model.py:10-11."""
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained(language_model_name, **huggingface_kwargs)


class _StepLoss(torch.autograd.Function):
    """training_step's loss: the kernels have already produced the gradients; backward() hands them to autograd."""

    @staticmethod
    def forward(ctx, model, loss, *params):
        ctx.model = model
        return loss.detach().clone()

    @staticmethod
    def backward(ctx, gout):
        model = ctx.model
        outs = []
        mg = model.transformer_mapper.engine.views(model.transformer_mapper.engine.arena.grads())
        lg = model.language_model.engine.views(model.language_model.engine.arena.grads()) if model._train_lm else {}
        sc = model.engine.scaler          # fp16 operands: the arenas hold loss-scale x gradient
        if sc is not None:
            gout = gout / sc.scale
        for owner, name in model._param_index:
            src = mg if owner == 0 else lg
            outs.append(src[name] * gout if name in src else None)
        return (None, None) + tuple(outs)


class ClipCapModel(nn.Module):
    _train_lm = True

    def __init__(self, config: Config, language_model: Optional[GPT2LM] = None):
        super().__init__()
        self.config = config
        self.hparams = config.to_dict()                      # what Lightning's save_hyperparameters exposed (model.py:16)
        self.language_model = language_model if language_model is not None else GPT2LM.from_pretrained(config.language_model)
        self.lm_embedding_size = self.language_model.get_input_embeddings().weight.shape[1]
        enc = config.encoder_config
        common = dict(encoder_embedding_size=enc.encoder_embedding_size, lm_embedding_size=self.lm_embedding_size,
                      prefix_length=config.prefix_length, projection_length=config.projection_length,
                      num_heads=config.transformer_attention_heads, num_layers=config.transformer_layers)
        if enc.use_windowed_embeddings:                      # model.py:22-32
            self.transformer_mapper = TransformerMapperWindowed(window_size=enc.window_size + 1,
                                                                use_pos_embeddings=config.use_positional_embeddings, **common)
        else:                                                # model.py:33-41
            self.transformer_mapper = TransformerMapper(**common)
        self._engine: Optional[ClipCapEngine] = None
        self._opt_step = 0
        self._param_index = [(0, n) for n in self.transformer_mapper._arena_params] + \
                            [(1, n) for n in self.language_model._arena_params]

    # ---- engine / fused path ----
    @property
    def engine(self) -> ClipCapEngine:
        if self._engine is None or self._engine.mapper is not self.transformer_mapper.engine:
            self._engine = ClipCapEngine(self.transformer_mapper.engine, self.language_model.engine, train_lm=self._train_lm)
        return self._engine

    def _dropout(self):
        """(p_embd, p_attn, p_resid, seed) for this step, or None.  The reference's ClipCapModel leaves the HF GPT-2 in train mode
        during a full finetune (model.py:19; only ClipCapModelPrefixOnly pins it to eval, :120-123), so its embd / attention /
        residual dropout is active; the seed comes from torch's CPU generator (torch.manual_seed reproduces a run) plus the rank."""
        lm = self.language_model
        if not (self._train_lm and lm.training):
            return None
        c = lm.config
        ps = (float(getattr(c, "embd_pdrop", 0.0)), float(getattr(c, "attn_pdrop", 0.0)), float(getattr(c, "resid_pdrop", 0.0)))
        if max(ps) <= 0.0:
            return None
        rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
        return ps + (int(torch.randint(0, 2 ** 48, (1,)).item()) ^ (rank << 50),)

    def fused_step(self, batch: Tuple[torch.Tensor, torch.Tensor], lr: float, reducer=None) -> torch.Tensor:
        """One optimizer step: zero grads -> forward+backward kernel chains -> (all-reduce) -> fused AdamW. Returns the loss."""
        tokens, embeds = batch
        eng = self.engine
        eng.zero_grad()
        if reducer is not None:
            reducer.begin()
            loss = eng.forward_backward(tokens, embeds, reduce_stats=reducer.reduce_stats, on_grads_ready=reducer.on_grads_ready,
                                        dropout=self._dropout())
            reducer.finish()
        else:
            loss = eng.forward_backward(tokens, embeds, dropout=self._dropout())
        self._opt_step += 1
        eng.optimizer_step(lr, self._opt_step, weight_decay=self._weight_decay(), sync_flag=(reducer.reduce_flag if reducer is not None else None))
        return loss

    def _weight_decay(self) -> float:
        """model.py:72-77: ``--enable-deepspeed`` swaps torch.optim.AdamW (weight_decay 0.01) for DeepSpeed's
        FusedAdam(adam_w_mode=True), whose default weight_decay is 0.0 — a run with that flag trains without decay; followed here."""
        tc = self.config.training_config
        return 0.0 if (tc is not None and tc.use_deepspeed_optimisers) else 0.01

    def set_precision(self, precision) -> "ClipCapModel":
        """What the reference hands to ``pl.Trainer(precision=args.fp_precision)`` (clipcap/train/train.py:82): 16 selects fp16 MFMA
        operands + dynamic loss scaling for mapper and language model, 32 (the reference's default) / 64 split-bf16 operands (three
        MFMA terms per product, logits within 1e-3 of the fp32 reference), "bf16" plain bf16 operands (_lib.op_dtype_of)."""
        self.transformer_mapper.set_precision(precision)
        self.language_model.set_precision(precision)
        self._engine = None
        return self

    # ---- reference surface ----
    def forward(self, tokens: torch.Tensor, embeddings: torch.Tensor, mask: Optional[torch.Tensor] = None):
        token_embeddings = self.language_model.get_input_embeddings()(tokens.clamp_min(0))
        prefix = self.transformer_mapper(embeddings)
        inputs_embeds = torch.cat((prefix, token_embeddings.to(prefix.device)), dim=1)
        return self.language_model(inputs_embeds=inputs_embeds, attention_mask=None)

    def set_training_config(self, training_config: TrainingConfig, reinit_optims: bool = False) -> None:
        self.config.training_config = training_config
        self.hparams = self.config.to_dict()
        if reinit_optims:
            self.configure_optimizers()

    def configure_optimizers(self) -> dict:
        tc = self.config.training_config
        assert tc is not None, "You must first use `set_training_config` before training."
        arenas = [self.transformer_mapper] + ([self.language_model] if self._train_lm else [])
        optimizer = ArenaAdamW(arenas, lr=tc.optimizer_lr, weight_decay=self._weight_decay())
        scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, linear_warmup_decay(tc.scheduler_warmup_steps, tc.total_steps))
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": scheduler, "interval": "step", "frequency": 1}}

    def training_step(self, batch: Tuple[torch.Tensor, torch.Tensor], _: int = 0) -> torch.Tensor:
        tokens, embeds = batch
        eng = self.engine
        eng.zero_grad()
        loss = eng.forward_backward(tokens, embeds, dropout=self._dropout())
        self.last_loss = loss
        if torch.is_grad_enabled():
            return _StepLoss.apply(self, loss, *[self._lookup(o, n) for o, n in self._param_index])
        return loss

    def _lookup(self, owner: int, name: str):
        return (self.transformer_mapper if owner == 0 else self.language_model)._arena_params[name]

    def log(self, *a, **k):   # Lightning API used by the reference's training_step (model.py:111); a no-op here
        pass


class ClipCapModelPrefixOnly(ClipCapModel):
    _train_lm = False

    def parameters(self, recurse: bool = True):
        return self.transformer_mapper.parameters()

    def train(self, mode: bool = True):
        super().train(mode)
        self.language_model.eval()
        return self
