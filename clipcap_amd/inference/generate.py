"""``generate`` — bos/text-prefix + mapper + sampling, the reference's clipcap/inference/generate.py:8-44."""
from __future__ import annotations

from typing import Callable, Optional

import torch

from clipcap_amd.inference.no_beam import generate_no_beam


def generate(model, tokenizer: Callable, embeddings: torch.Tensor, top_p: float = 0.95, top_k: int = 0, temperature: float = 1.0,
             number_to_generate: int = 5, text_prefix: Optional[str] = None, stop_token: Optional[str] = None,
             generator: Optional[torch.Generator] = None):
    batch_size = embeddings.shape[0]
    assert batch_size == 1, "Batch size > 1 support coming soon - for now leave embeddings.shape[0] as 1."   # generate.py:19-20
    text_prefix = tokenizer.bos_token + (text_prefix or "")                                                  # generate.py:22-25
    text_prefix_tokens = tokenizer.encode(text_prefix, return_tensors="pt").expand(batch_size, -1).to(embeddings.device)
    with torch.no_grad():
        token_embeddings = model.language_model.get_input_embeddings()(text_prefix_tokens)
        prefix_projections = model.transformer_mapper(embeddings)
    inputs_embeds = torch.cat((prefix_projections, token_embeddings.to(prefix_projections.device)), dim=1)
    # like the reference, generate_no_beam appends the text prefix's embeddings AGAIN (it is given text_prefix_tokens, no_beam.py:28-30):
    # the language model sees [prefix ; bos+text ; bos+text ; generated...]
    return generate_no_beam(model, tokenizer, inputs_embeds, number_to_generate=number_to_generate, text_prefix_tokens=text_prefix_tokens,
                            top_p=top_p, top_k=top_k, temperature=temperature, generator=generator)
