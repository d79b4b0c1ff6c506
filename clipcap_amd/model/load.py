"""``load(model_path, config_path, device, from_checkpoint)`` -> (model.eval(), tokenizer) — reference clipcap/model/load.py:9-42.
Reads the same ``*_config.yaml`` schema and the same state-dict keys (``strict=False``), so reference checkpoints load."""
from __future__ import annotations

from typing import Callable, Tuple, Union

import torch
import yaml

from clipcap_amd.encoders.config import EncoderConfig
from clipcap_amd.model.config import Config
from clipcap_amd.model.model import ClipCapModel, ClipCapModelPrefixOnly, get_tokenizer


def load(model_path: str, config_path: str, device: str = "cpu", from_checkpoint: bool = False,
         tokenizer: Union[None, Callable] = None) -> Tuple[Union[ClipCapModel, ClipCapModelPrefixOnly], Callable]:
    with open(config_path, "r") as f:
        raw = yaml.safe_load(f)
    if from_checkpoint and raw.get("training_config") is not None:
        raw["training_config"] = None                       # stale schedule of a finished run (load.py:15-16)
    raw["encoder_config"] = EncoderConfig(**raw["encoder_config"])
    if isinstance(raw.get("training_config"), dict):
        from clipcap_amd.model.config import TrainingConfig
        raw["training_config"] = TrainingConfig(**raw["training_config"])
    config = Config(**raw)
    model = (ClipCapModel if config.train_language_model else ClipCapModelPrefixOnly)(config)
    state = torch.load(model_path, map_location="cpu")
    if from_checkpoint:
        state = state["state_dict"]
    model.load_state_dict(state, strict=False)
    model = model.eval().to(device)
    if tokenizer is None:
        tokenizer = get_tokenizer(config.language_model)
    return model, tokenizer
