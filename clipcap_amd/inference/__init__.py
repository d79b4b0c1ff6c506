"""clipcap_amd.inference — decoding entry points (reference clipcap/inference/)."""
from clipcap_amd.inference.base import generate_beam, generate_beam_tokens, generate_no_beam, generate_nucleus_sampling  # noqa: F401
from clipcap_amd.inference.generate import generate  # noqa: F401
