"""``generate_nucleus_sampling`` of the reference's clipcap/inference/nucleus_sampling.py:9-74.

Same sampling rule as inference/base.py:135-201 (top-k of the softmax, minimal prefix with cumulative mass >= top_p relative to
the FULL softmax, renormalise) but it stops on the token of ``"."`` (:21) instead of eos, and ``top_k`` is an int whose 0 means
"all" (:40-41).  The returned text includes the stop token (appended before the break, :61-69).  KV-cached batched decode with
cc_sample_step mode 0, every repetition of ``number_to_generate`` one row.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch

from clipcap_amd.inference.base import _rows_for, _with_text_prefix, sample_tokens


def generate_nucleus_sampling(model, tokenizer: Callable, embeds: torch.Tensor, number_to_generate: int = 1,
                              text_prefix_tokens: Optional[torch.Tensor] = None, entry_length: int = 67, top_p: float = 0.8,
                              top_k: int = 0, temperature: float = 1.0, generator: Optional[torch.Generator] = None) -> List[str]:
    stop = tokenizer.encode(".")[0]                                                   # nucleus_sampling.py:21
    embeds = _with_text_prefix(model, embeds, text_prefix_tokens)
    toks, stop_pos = sample_tokens(model, _rows_for(embeds, number_to_generate), entry_length, stop, mode=0,
                                   top_p=1.0 if top_p is None else top_p, top_k=int(top_k or 0), temperature=temperature, generator=generator)
    head = [] if text_prefix_tokens is None else [int(t) for t in text_prefix_tokens.flatten()]
    toks, stop_pos = toks.cpu(), stop_pos.cpu()
    return [tokenizer.decode(head + toks[r, :min(int(stop_pos[r]) + 1, toks.shape[1])].tolist()) for r in range(toks.shape[0])]
