"""Encoder entry points of the reference's surface (clipcap/encoders/base.py:10-46) — as HOOKS.

The CLIP / CLAP encoders are out of scope here (they stay upstream, frozen, and hand over precomputed embeddings
through the embedding-reader path), but pipelines written against the reference call ``clipcap.get_encoder_from_model(model, device)``
and expect ``(encode_fn, preprocess)`` back.  An upstream encoder is plugged in once with ``register_encoder(name, factory)``; the three
reference functions then resolve it with the reference's own argument meaning.  Nothing is registered by default, and asking for an
unregistered encoder raises with the name that is missing (the reference raises ValueError for an unknown name, base.py:24-26).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

from clipcap_amd.encoders.config import EncoderConfig

_FACTORIES: Dict[str, Callable] = {}


def register_encoder(name: str, factory: Callable) -> None:
    """``factory(encoder_model_variant, normalize_embeddings=..., window_size=..., use_windowed_embeddings=...,
    window_overlap_percentage=..., device=...) -> (encode_fn, preprocess)`` — the keyword set of get_encoder below."""
    _FACTORIES[name] = factory


def get_encoder(encoder_model_name: str, encoder_model_variant: str, normalize_embeddings: bool = False, window_size: Optional[int] = None,
                use_windowed_embeddings: bool = False, window_overlap_percentage: float = 0.0, device: str = "cuda") -> Tuple[Callable, Callable]:
    """base.py:10-26."""
    if encoder_model_name not in _FACTORIES:
        raise ValueError(f"invalid encoder name: '{encoder_model_name}' — no upstream encoder is registered under it "
                         f"(clipcap_amd keeps CLIP / CLAP upstream: clipcap_amd.encoders.register_encoder(name, factory), or feed "
                         f"precomputed embeddings to model.transformer_mapper)")
    return _FACTORIES[encoder_model_name](encoder_model_variant, normalize_embeddings=normalize_embeddings, window_size=window_size,
                                          use_windowed_embeddings=use_windowed_embeddings,
                                          window_overlap_percentage=window_overlap_percentage, device=device)


def get_encoder_from_config(config: EncoderConfig, device: str = "cpu") -> Tuple[Callable, Callable]:
    """base.py:29-38 (the CLIP variant is stored with '_' for '/' in the yaml)."""
    variant = config.encoder_model_variant
    if config.encoder_model_name == "clip" and isinstance(variant, str):
        variant = variant.replace("_", "/")
    return get_encoder(config.encoder_model_name, variant, normalize_embeddings=config.normalize_embeddings,
                       use_windowed_embeddings=config.use_windowed_embeddings, window_size=config.window_size,
                       window_overlap_percentage=config.window_overlap_percentage, device=device)


def get_encoder_from_model(model, device: str = "cpu") -> Tuple[Callable, Callable]:
    """base.py:40-41."""
    return get_encoder_from_config(model.config.encoder_config, device=device)
