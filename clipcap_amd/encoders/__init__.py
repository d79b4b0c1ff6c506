"""clipcap_amd.encoders — the encoder *configuration* and the encoder entry points as hooks (clipcap/encoders/__init__.py:1-3).
CLIP / CLAP themselves stay upstream as precomputed embeddings; see base.py."""
from clipcap_amd.encoders.base import get_encoder, get_encoder_from_config, get_encoder_from_model, register_encoder  # noqa: F401
from clipcap_amd.encoders.config import EncoderConfig  # noqa: F401
