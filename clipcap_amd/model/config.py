"""Config / TrainingConfig — same fields, defaults and yaml schema as the reference (clipcap/model/config.py:7-55), so a
``*_config.yaml`` written by either implementation loads in the other."""
from __future__ import annotations

from argparse import Namespace
from dataclasses import asdict, dataclass
from typing import Optional

from clipcap_amd.encoders.config import EncoderConfig


@dataclass
class TrainingConfig:
    optimizer_lr: float = 2e-5
    use_deepspeed_optimisers: bool = True   # the fused HIP AdamW is always used; True selects FusedAdam's weight decay (0.0), False torch AdamW's (0.01): model.py:72-77
    scheduler_warmup_steps: int = 123
    total_steps: int = 123

    def to_dict(self) -> dict:
        return asdict(self)

    @classmethod
    def from_args(cls, args: Namespace) -> "TrainingConfig":
        return cls(optimizer_lr=args.optimizer_lr, use_deepspeed_optimisers=args.enable_deepspeed,
                   scheduler_warmup_steps=args.scheduler_warmup_steps, total_steps=args.total_steps)


@dataclass
class Config:
    language_model: str = "gpt2-xl"
    train_language_model: bool = False
    prefix_length: int = 10
    projection_length: int = 10
    transformer_layers: int = 8
    transformer_attention_heads: int = 16
    use_positional_embeddings: bool = True
    encoder_config: Optional[EncoderConfig] = None
    training_config: Optional[TrainingConfig] = None

    def to_dict(self) -> dict:
        return asdict(self)

    @classmethod
    def from_args(cls, args: Namespace) -> "Config":
        names = ("language_model", "train_language_model", "prefix_length", "projection_length", "transformer_layers",
                 "transformer_attention_heads", "use_positional_embeddings")
        return cls(encoder_config=None, training_config=None, **{n: getattr(args, n) for n in names})
