"""clipcap_amd — MI355X-native ClipCap training + caption-generation path (HIP kernels behind a C ABI).

Mirrors the reference's import surface (clipcap/__init__.py:1-2): ``clipcap_amd.load`` plus the
``clipcap_amd.model`` / ``clipcap_amd.train`` / ``clipcap_amd.inference`` sub-packages.
"""
__version__ = "0.1.0"


def load(*args, **kwargs):
    """clipcap.load (reference clipcap/model/load.py:9-42)."""
    from clipcap_amd.model.load import load as _load
    return _load(*args, **kwargs)
