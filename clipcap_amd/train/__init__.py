"""clipcap_amd.train — mirrors clipcap/train/__init__.py:1-2 (``train``, ``start_training``, ``add_training_args``)."""
from clipcap_amd.train.args import add_training_args  # noqa: F401
from clipcap_amd.train.train import start_training, train  # noqa: F401
