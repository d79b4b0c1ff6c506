"""Module alias kept for import compatibility with clipcap/inference/nucleus_sampling.py (the reference keeps near-duplicate copies of its
decode loops there; here there is one implementation in clipcap_amd.inference.base)."""
from clipcap_amd.inference.base import generate_no_beam, generate_nucleus_sampling  # noqa: F401
