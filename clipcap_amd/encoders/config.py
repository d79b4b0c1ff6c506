"""EncoderConfig — field-for-field the schema of the reference's ``encoder_config.yaml``
(reference clipcap/encoders/config.py:6-28; written by preprocess, read by train.py:26-29 and load.py:21)."""
from __future__ import annotations

from argparse import Namespace
from dataclasses import asdict, dataclass
from typing import Optional


@dataclass
class EncoderConfig:
    encoder_model_name: str = "clip"
    encoder_model_variant: str = "ViT-L/14"
    encoder_embedding_size: Optional[int] = None   # filled in by the dataloader (train.py:39)
    normalize_embeddings: bool = False
    use_windowed_embeddings: bool = False
    window_size: int = 16
    window_overlap_percentage: float = 0.0

    def to_dict(self) -> dict:
        return asdict(self)

    @classmethod
    def from_args(cls, args: Namespace) -> "EncoderConfig":
        names = ("encoder_model_name", "encoder_model_variant", "normalize_embeddings", "use_windowed_embeddings", "window_size",
                 "window_overlap_percentage")
        return cls(encoder_embedding_size=None, **{n: getattr(args, n) for n in names})
