"""clipcap_amd.model — same public names as clipcap/model/__init__.py:1-4."""
from clipcap_amd.model.args import add_model_args  # noqa: F401
from clipcap_amd.model.config import Config, TrainingConfig  # noqa: F401
from clipcap_amd.model.load import load  # noqa: F401
from clipcap_amd.model.model import ClipCapModel, ClipCapModelPrefixOnly, get_tokenizer  # noqa: F401
