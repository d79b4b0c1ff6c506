"""Logit post-processing used by the sampling decoders — same names/semantics as the reference's helpers
(clipcap/inference/utils.py:5-51, duplicated in inference/base.py:9-55).  Unlike the reference these work on device tensors
of any batch shape (..., V) and do not modify their input."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def top_k_top_p_filtering(logits: torch.Tensor, top_k: int = 0, top_p: float = 0.0, filter_value: float = -float("inf")) -> torch.Tensor:
    """Keep the top_k highest logits and/or the smallest sorted set whose cumulative probability exceeds top_p (the first
    token above the threshold is kept, utils.py:25-28)."""
    out = logits.clone()
    k = min(int(top_k), out.size(-1))
    if k > 0:
        kth = torch.topk(out, k, dim=-1).values[..., -1:]
        out = out.masked_fill(out < kth, filter_value)
    if top_p > 0.0:
        sorted_logits, sorted_idx = torch.sort(out, descending=True, dim=-1)
        cum = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
        drop_sorted = cum > top_p
        drop_sorted = torch.cat((torch.zeros_like(drop_sorted[..., :1]), drop_sorted[..., :-1]), dim=-1)
        drop = torch.zeros_like(drop_sorted).scatter(-1, sorted_idx, drop_sorted)
        out = out.masked_fill(drop, filter_value)
    return out


def repetition_penalty_apply(logits: torch.Tensor, tokens: torch.Tensor, penalty: float) -> torch.Tensor:
    """utils.py:33-37: already-generated tokens get logit*penalty if negative else logit/penalty."""
    out = logits.clone()
    t = torch.gather(out, -1, tokens)
    return out.scatter(-1, tokens, torch.where(t < 0, t * penalty, t / penalty))


def sentence_length_penalty_apply(logits: torch.Tensor, tokens: torch.Tensor, stop_token: int, current_length: int, desired_length: int,
                                  length_factor: float) -> torch.Tensor:
    """utils.py:39-49.  NB: like the reference this compares the gathered logit VALUES with ``stop_token`` (not token ids)."""
    out = logits.clone()
    penalty = (current_length / desired_length) * length_factor
    t = torch.gather(out, -1, tokens)
    return out.scatter(-1, tokens, torch.where(t == stop_token, t * penalty, t))


def nucleus_distribution(logits: torch.Tensor, top_p: float = 0.8, top_k=None) -> torch.Tensor:
    """Pre-sampling distribution of generate_nucleus_sampling (inference/base.py:165-181): probabilities sorted descending,
    cut at the first cumulative mass >= top_p (inclusive), renormalised, scattered back.  logits (n, V) -> (n, V)."""
    V = logits.shape[-1]
    top_k = V if top_k is None else top_k
    top_p = 1.0 if top_p is None else top_p
    p, idx = F.softmax(logits, dim=-1).topk(top_k, dim=-1)
    cum = p.cumsum(dim=-1)
    thr = torch.full((p.shape[0], 1), float(top_p), device=logits.device, dtype=cum.dtype)
    cut_i = torch.searchsorted(cum, thr).clamp(max=top_k - 1)
    cut = torch.gather(cum, -1, cut_i)
    kept = (cum <= cut) * p
    kept = kept / kept.sum(dim=-1, keepdim=True)
    return torch.zeros_like(logits).scatter(-1, idx, kept.to(logits.dtype))
