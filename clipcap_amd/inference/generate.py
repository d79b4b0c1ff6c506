"""``generate`` — bos/text-prefix + mapper + sampling, the reference's clipcap/inference/generate.py:8-44."""
from __future__ import annotations

from typing import Callable, Optional

import torch

from clipcap_amd.inference.base import generate_no_beam


def generate(model, tokenizer: Callable, embeddings: torch.Tensor, top_p: float = 0.95, top_k: int = 0, temperature: float = 1.0,
             number_to_generate: int = 5, text_prefix: Optional[str] = None, stop_token: Optional[str] = None):
    assert embeddings.shape[0] == 1, "batch-1 like the reference (generate.py:19-20); use generate_beam for batches"
    text = tokenizer.bos_token + (text_prefix or "")
    prefix_tokens = tokenizer.encode(text, return_tensors="pt").to(embeddings.device)
    with torch.no_grad():
        tok_emb = model.language_model.get_input_embeddings()(prefix_tokens)
        prefix = model.transformer_mapper(embeddings)
    inputs = torch.cat((prefix, tok_emb.to(prefix.device)), dim=1)
    outs = []
    for _ in range(number_to_generate):
        outs += generate_no_beam(model, tokenizer, inputs, top_p=top_p, top_k=top_k, temperature=temperature, sweep=False)
    return outs
