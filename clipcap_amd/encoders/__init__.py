"""Only the encoder *configuration* is in scope: CLIP/CLAP stay upstream as precomputed embeddings."""
from clipcap_amd.encoders.config import EncoderConfig  # noqa: F401
