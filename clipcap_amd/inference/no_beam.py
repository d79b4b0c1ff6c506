"""``generate_no_beam`` of the reference's clipcap/inference/no_beam.py:10-82 — the variant ``generate()`` calls (generate.py:34-41).

It is NOT the function of the same name in inference/base.py (:204-279, a top_p x temperature debugging sweep): this one
  * stops on the token of ``"."`` (no_beam.py:24), not on eos;
  * samples ``number_to_generate`` captions (no_beam.py:33);
  * applies the repetition penalty (default 1.2) to every token already in ``tokens`` = text_prefix_tokens ++ generated
    (no_beam.py:45-48; utils.py:33-37), BEFORE temperature and top-k / top-p filtering (:50-52);
  * returns text_prefix_tokens ++ generated without the stop token (it breaks before appending, :67-75).
  * applies the sentence-length penalty (no_beam.py:55-60) the way the reference does: after the filter, history tokens whose logit
    VALUE equals the stop-token id (utils.py:45 compares gathered logit values with the id) are multiplied by
    len(tokens) / desired_sentence_length * sentence_length_factor.

Device path: one KV-cached batched decode (cc_decode_fwd) with all ``number_to_generate`` repetitions as rows, the whole
per-step rule (penalties with a history bitmap, temperature, filter, softmax, draw) in cc_sample_step_lp.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch

from clipcap_amd.inference.base import _rows_for, _with_text_prefix, sample_tokens


def generate_no_beam(model, tokenizer: Callable, embeds: torch.Tensor, number_to_generate: int = 5,
                     text_prefix_tokens: Optional[torch.Tensor] = None, top_p: float = 0.9, top_k: float = 0.0, entry_length: int = 67,
                     temperature: float = 1.0, repetition_penalty: float = 1.2, desired_sentence_length: int = 50,
                     sentence_length_factor: float = 1.0, generator: Optional[torch.Generator] = None) -> List[str]:
    stop = tokenizer.encode(".")[0]                                                   # no_beam.py:24
    embeds = _with_text_prefix(model, embeds, text_prefix_tokens)                     # no_beam.py:28-30
    rows = _rows_for(embeds, number_to_generate)
    toks, stop_pos = sample_tokens(model, rows, entry_length, stop, mode=1, top_p=top_p, top_k=int(top_k), temperature=temperature,
                                   repetition_penalty=repetition_penalty, generator=generator, head_tokens=text_prefix_tokens,
                                   desired_sentence_length=desired_sentence_length, sentence_length_factor=sentence_length_factor)
    head = [] if text_prefix_tokens is None else [int(v) for v in text_prefix_tokens.flatten()]
    toks, stop_pos = toks.cpu(), stop_pos.cpu()
    return [tokenizer.decode(head + toks[r, :int(stop_pos[r])].tolist()) for r in range(toks.shape[0])]
