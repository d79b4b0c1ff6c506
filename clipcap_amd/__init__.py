"""clipcap_amd — MI355X-native ClipCap training + caption-generation path (HIP kernels behind a C ABI).

Mirrors the reference's import surface (clipcap/__init__.py:1-2): ``clipcap_amd.load``, ``get_encoder`` / ``get_encoder_from_model`` and the
``clipcap_amd.model`` / ``clipcap_amd.train`` / ``clipcap_amd.inference`` / ``clipcap_amd.encoders`` sub-packages.
``install_as_clipcap()`` registers the same modules under the reference's own name so that ``import clipcap`` / ``python -m clipcap.train``
call sites need no edit.
"""
import importlib
import sys

__version__ = "0.2.0"

# modules of the reference's package that have a counterpart here (clipcap/<name>.py -> clipcap_amd/<name>.py)
_ALIASED = ("model", "model.model", "model.config", "model.args", "model.load", "model.mapper", "train", "train.train", "train.args",
            "train.callback", "train.dataloader", "train.__main__", "inference", "inference.base", "inference.generate", "inference.no_beam",
            "inference.nucleus_sampling", "inference.utils", "encoders", "encoders.base", "encoders.config")


def load(*args, **kwargs):
    """clipcap.load (reference clipcap/model/load.py:9-42)."""
    from clipcap_amd.model.load import load as _load
    return _load(*args, **kwargs)


def get_encoder(*args, **kwargs):
    """clipcap.get_encoder (reference clipcap/encoders/base.py:10-26) — resolves an encoder registered with clipcap_amd.encoders.register_encoder."""
    from clipcap_amd.encoders.base import get_encoder as _g
    return _g(*args, **kwargs)


def get_encoder_from_model(*args, **kwargs):
    """clipcap.get_encoder_from_model (reference clipcap/encoders/base.py:40-41)."""
    from clipcap_amd.encoders.base import get_encoder_from_model as _g
    return _g(*args, **kwargs)


def install_as_clipcap(force: bool = False) -> None:
    """Makes ``import clipcap``, ``clipcap.model``, ``clipcap.train``, ``clipcap.inference[.base|.generate|.no_beam|.nucleus_sampling]`` and
    ``clipcap.encoders[.config]`` resolve to this package (entries in ``sys.modules``; ``python -m clipcap_amd.train`` stays the CLI, and after
    the call ``runpy.run_module("clipcap.train")`` works too).  Refuses when a different ``clipcap`` package is already imported —
    silently shadowing the reference inside a process that uses it would mix the two — unless ``force`` is set.  Idempotent."""
    me = sys.modules[__name__]
    cur = sys.modules.get("clipcap")
    if cur is not None and cur is not me and not force:
        raise RuntimeError(f"a different 'clipcap' is already imported ({getattr(cur, '__file__', cur)!r}); "
                           "call clipcap_amd.install_as_clipcap() before importing it, or pass force=True")
    if cur is not None and cur is not me:
        for k in [k for k in sys.modules if k == "clipcap" or k.startswith("clipcap.")]:
            del sys.modules[k]
    sys.modules["clipcap"] = me
    for name in _ALIASED:
        if name.endswith("__main__"):
            continue                                      # never imported ahead of time: importing it would run the CLI
        sys.modules["clipcap." + name] = importlib.import_module("clipcap_amd." + name)
