"""Caption decoding with a KV cache on the HIP path — same entry points as the reference's clipcap/inference/base.py
(generate_beam :55-132, generate_nucleus_sampling :135-201, generate_no_beam :204-279) with the same arguments.

Differences that do not change results: GPT-2 runs incrementally (cc_decode_fwd) instead of re-forwarding the growing
sequence every step (base.py:81), the beam update runs in one device kernel (cc_beam_step), and ``embeds`` may carry a
batch of prefixes (B, L, D): the reference is batch-1 only, and B == 1 reproduces its outputs.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple, Union

import torch

from clipcap_amd.engine import DecodeSession, beam_buffers, beam_step, embed_tokens, sample_step
from clipcap_amd.inference.utils import (nucleus_distribution, repetition_penalty_apply, sentence_length_penalty_apply,  # noqa: F401
                                         top_k_top_p_filtering)


def _persistent_decode_on() -> bool:
    from clipcap_amd import _lib
    return bool(_lib.lib().cc_decode_mode(-1) & 6)          # bit 1: persistent layer launch, bit 2: XCD-team engine (include/clipcap_hip.h)


def _stop_id(tokenizer) -> int:
    return tokenizer.encode(tokenizer.eos_token)[0]          # base.py:66


def _with_text_prefix(model, embeds, text_prefix_tokens):
    if text_prefix_tokens is None:
        return embeds
    emb = model.language_model.get_input_embeddings()(text_prefix_tokens.to(embeds.device))
    return torch.cat((embeds, emb.expand(embeds.shape[0], -1, -1)), dim=1)   # base.py:75-77


@torch.no_grad()
def generate_beam_rounds(model, embeds: torch.Tensor, beam_size: int = 5, entry_length: int = 67, temperature: float = 1.0,
                         stop_token: int = 50256, rounds: int = 1) -> List[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
    """Token-level beam search for a batch of prefixes, embeds fp32 (S, L, D), as ``rounds`` consecutive generations with LIVE state —
    the reference's ``for _ in range(number_to_generate)`` around its entry_length loop (base.py:79-130): a later round continues the beams
    of the one before (token histories, has_stopped, seq_lengths, the KV cache), its running ``scores`` are the previous round's scores
    already divided by seq_lengths (base.py:123 rebinds the name), beams that never stopped keep generating for another entry_length
    tokens, and a round that starts with every beam stopped still takes one step (base.py:80-121) before it breaks.
    Returns one (tokens int64 (S, beam, n_r), scores (S, beam) length-normalised, seq_lengths (S, beam)) per round."""
    lm = model.language_model
    g = lm.engine
    dev = g.arena.device
    embeds = embeds.to(dev, torch.float32)
    S, L0, D = embeds.shape
    R = S * beam_size
    V = g.dims["V"]
    rounds = max(1, int(rounds))
    # positions are allocated up to n_positions; like the reference (base.py:79-130), a large number_to_generate only fails if generation
    # actually REACHES the position limit — beams normally stop far earlier
    total = min(rounds * entry_length, g.dims["NPOS"] - L0)
    if total < 1:
        raise RuntimeError(f"a {L0}-position prefix leaves no room to generate (n_positions = {g.dims['NPOS']})")
    wte = lm.get_input_embeddings().weight.detach()
    scores = torch.zeros(R, dtype=torch.float32, device=dev)
    seq_lengths = torch.ones(R, dtype=torch.float32, device=dev)
    has_stopped = torch.zeros(R, dtype=torch.uint8, device=dev)
    base = (torch.arange(S, device=dev, dtype=torch.int32) * beam_size).repeat_interleave(beam_size)
    # step 0: one row per sample (base.py:86-94), then fan the cache out to beam rows
    sess = DecodeSession(g, S, L0 + total)
    logits0 = sess.forward(embeds)                                              # (S, V)
    lg = torch.empty(R, V, dtype=torch.float32, device=dev)
    lg[::beam_size] = logits0                                                   # row 0 of every beam set
    bufs = beam_buffers(dev, S, beam_size, V)                                   # this decode's own step outputs + scratch
    next_tok, src = beam_step(lg, S, beam_size, temperature, True, stop_token, scores, seq_lengths, has_stopped, bufs)
    sess = sess.expand((base // beam_size).to(torch.int32), R)
    # token histories (int32, ping-pong), the next input embedding and the cache ancestry are advanced by one launch per step
    tok = [torch.zeros(R, total, dtype=torch.int32, device=dev) for _ in range(2)]
    x = torch.empty(R, 1, D, dtype=torch.float32, device=dev)
    sess.beam_advance(beam_size, next_tok, None, wte, 0, tok[1], tok[0], x)     # step 0: tokens = next_tokens (base.py:94)
    n = 1                                                                       # token columns written so far
    out = []
    for r in range(rounds):
        for i in range(1 if r == 0 else 0, entry_length):
            # base.py:120-121 breaks as soon as every beam has stopped.  Steps taken after that point only append token 0 to frozen
            # beams (lengths and the truncated outputs are unchanged; scores are re-sorted and change by ulps at most), so polling the flag every 4th step — one host sync
            # instead of four — cannot change the result.  (i == 0 of a later round is never skipped: the reference takes that step.)
            if i % 4 == 1 and bool(has_stopped.all()):
                break
            if n >= total:
                raise RuntimeError(f"generation reached n_positions = {g.dims['NPOS']} ({L0}-position prefix + {n} tokens; round {r + 1} of {rounds})")
            logits = sess.forward(x, partials=True, group=beam_size)            # x = wte[next_tokens] (base.py:117)
            next_tok, src = beam_step(logits, S, beam_size, temperature, False, stop_token, scores, seq_lengths, has_stopped, bufs, sess.lpart)
            sess.beam_advance(beam_size, next_tok, src, wte, n, tok[(n - 1) & 1], tok[n & 1], x)   # base.py:104-117
            n += 1
        final = scores / seq_lengths                                            # base.py:123
        out.append((tok[(n - 1) & 1][:, :n].to(torch.int64).view(S, beam_size, -1), final.view(S, beam_size).clone(),
                    seq_lengths.view(S, beam_size).clone()))
        scores.copy_(final)                                                     # the next round starts from the normalised scores (base.py:123)
    if _persistent_decode_on():
        sess.check()                                                            # a persistent decode launch that gave up must not go unnoticed (the flag is sticky)
    return out


def generate_beam_tokens(model, embeds: torch.Tensor, beam_size: int = 5, entry_length: int = 67, temperature: float = 1.0,
                         stop_token: int = 50256) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """One generation (the reference's number_to_generate = 1): (tokens int64 (S, beam, n), scores (S, beam) length-normalised,
    seq_lengths (S, beam))."""
    return generate_beam_rounds(model, embeds, beam_size, entry_length, temperature, stop_token, 1)[0]


def generate_beam(model, tokenizer: Callable, embeds: torch.Tensor, number_to_generate: int = 1,
                  text_prefix_tokens: Optional[torch.Tensor] = None, beam_size: int = 5, entry_length: int = 67,
                  temperature: float = 1.0) -> List[str]:
    """Reference signature (base.py:55-64).  For the reference's batch-1 input: ``number_to_generate`` texts, the best beam after each of
    the consecutive generations (base.py:79-130, see generate_beam_rounds).  A batch of prefixes (S > 1, an extension — the reference is
    batch-1) returns the best caption of every prefix row, round-major when number_to_generate > 1."""
    stop = _stop_id(tokenizer)
    embeds = _with_text_prefix(model, embeds, text_prefix_tokens)
    out = []
    for tokens, scores, lengths in generate_beam_rounds(model, embeds, beam_size, entry_length, temperature, stop, number_to_generate):
        best = scores.argmax(dim=1)     # == argsort(descending)[0]; ties resolve to the first beam like a stable sort
        for s in range(tokens.shape[0]):
            b = int(best[s])
            n = int(lengths[s, b])
            out.append(tokenizer.decode(tokens[s, b, :n].cpu().numpy()))
    return out


@torch.no_grad()
def sample_tokens(model, embeds: torch.Tensor, entry_length: int = 67, stop_token: int = 50256, *, mode: int = 0, top_p: float = 0.8,
                  top_k: Optional[int] = None, temperature: float = 1.0, repetition_penalty: float = 1.0,
                  generator: Optional[torch.Generator] = None, head_tokens: Optional[torch.Tensor] = None,
                  desired_sentence_length: Optional[int] = None, sentence_length_factor: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """KV-cached sampling for a batch of prefixes, every step on the device (cc_decode_fwd + cc_sample_step, no host sync except
    an all-rows-stopped poll every 4th step).  embeds fp32 (R, L, D).  mode 0 = the nucleus rule of generate_nucleus_sampling
    (base.py:165-181), mode 1 = top_k_top_p_filtering + softmax (base.py:245-262).
    ``head_tokens`` int64 (R, H0) or (1, H0): tokens that precede the generated ones in the repetition-penalty history (the
    reference penalises ``tokens`` = text_prefix_tokens ++ generated, no_beam.py:32,45-48).
    ``desired_sentence_length`` not None: the sentence-length penalty of no_beam.py:55-60 — with a non-empty history, history tokens
    whose filtered logit equals float(stop_token) are scaled by (len(history) / desired_sentence_length) * sentence_length_factor
    (utils.py:40-51) before the softmax, inside cc_sample_step_lp.
    Returns (tokens int64 (R, n), stop_pos int64 (R,)): stop_pos[r] = index of the first stop token of row r (n if none)."""
    lm = model.language_model
    g = lm.engine
    dev = g.arena.device
    embeds = embeds.to(dev, torch.float32)
    R, L0, D = embeds.shape
    sess = DecodeSession(g, R, L0 + entry_length)
    H0 = 0 if head_tokens is None else int(head_tokens.shape[-1])
    hist = torch.zeros(R, H0 + entry_length, dtype=torch.int64, device=dev)          # [head | generated]: the penalty's history
    if H0:
        hist[:, :H0] = head_tokens.to(dev, torch.int64).reshape(-1, H0).expand(R, H0)
    toks = hist[:, H0:]
    xbuf = torch.empty(R, 1, D, dtype=torch.float32, device=dev)
    x = embeds
    n = 0
    for step in range(entry_length):
        logits = sess.forward(x)                                                     # (R, V) fp32
        u = torch.rand(R, device=dev, generator=generator)
        nxt = sample_step(logits, u, temperature=temperature, top_k=top_k or 0, top_p=top_p, mode=mode, history=hist, hist_len=H0 + step,
                          repetition_penalty=repetition_penalty,
                          length_penalty_stop=stop_token if desired_sentence_length is not None else -1,
                          length_penalty=((H0 + step) / desired_sentence_length) * sentence_length_factor if desired_sentence_length else 1.0)  # int32 (R,)
        toks[:, step] = nxt
        n = step + 1
        # every row has produced its stop token: polled every 4th step (one host sync instead of four; the extra tokens are cut below)
        if step % 4 == 3 and bool((toks[:, :n] == stop_token).any(dim=1).all()):
            break
        x = embed_tokens(g, nxt, xbuf)                                               # wte[next] (base.py:184), one launch
    toks = toks[:, :n]
    hit = toks == stop_token
    stop_pos = torch.where(hit.any(dim=1), hit.to(torch.int32).argmax(dim=1), torch.full((R,), n, dtype=torch.int64, device=dev))
    return toks, stop_pos


def _rows_for(embeds: torch.Tensor, number_to_generate: int) -> torch.Tensor:
    return embeds.repeat_interleave(max(1, number_to_generate), dim=0) if number_to_generate > 1 else embeds


def generate_nucleus_sampling(model, tokenizer: Callable, embeds: torch.Tensor, number_to_generate: int = 1,
                              text_prefix_tokens: Optional[torch.Tensor] = None, entry_length: int = 67, top_p: float = 0.8, top_k=None,
                              temperature: float = 1.0, generator: Optional[torch.Generator] = None) -> List[str]:
    """base.py:135-201 with a KV cache and the sampling step on the device.  The reference is batch-1 and loops number_to_generate
    times; here every (prefix row, repetition) is one row of a single batched decode.  The returned text of a row includes its
    stop token, as the reference's does (it appends before breaking, base.py:186-195)."""
    stop = _stop_id(tokenizer)
    embeds = _with_text_prefix(model, embeds, text_prefix_tokens)
    toks, stop_pos = sample_tokens(model, _rows_for(embeds, number_to_generate), entry_length, stop, mode=0, top_p=top_p, top_k=top_k,
                                   temperature=temperature, generator=generator)
    head = [] if text_prefix_tokens is None else [int(t) for t in text_prefix_tokens.flatten()]
    toks, stop_pos = toks.cpu(), stop_pos.cpu()
    return [tokenizer.decode(head + toks[r, :min(int(stop_pos[r]) + 1, toks.shape[1])].tolist()) for r in range(toks.shape[0])]


def generate_no_beam(model, tokenizer: Callable, embeds: torch.Tensor, text_prefix_tokens: Optional[torch.Tensor] = None,
                     top_p: float = 0.9, top_k: float = 0.0, entry_length: int = 67, temperature: float = 1.0,
                     repetition_penalty: float = 1.2, desired_sentence_length: int = 50, sentence_length_factor: float = 1.0,
                     sweep: bool = True, generator: Optional[torch.Generator] = None) -> List[str]:
    """base.py:204-279: the reference's debugging sweep over top_p x temperature (sweep=True reproduces it; sweep=False samples
    once with the given top_p / temperature).  Like the reference, ``repetition_penalty`` is accepted but not applied (its call is
    commented out, base.py:236-239).  The sentence-length penalty (base.py:248-254) is applied as the reference applies it: after the
    filter, history tokens whose VALUE equals the stop token id (utils.py:45 compares values, not ids) are scaled — on the device,
    inside the sampling step.  The text of a row excludes its stop token (the reference breaks before appending, base.py:261-262)."""
    stop = _stop_id(tokenizer)
    embeds = _with_text_prefix(model, embeds, text_prefix_tokens)
    grid = [(p, t) for p in (0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95, 1.0) for t in (0.9, 0.95, 1.0)] if sweep \
        else [(top_p, temperature)]
    head = [] if text_prefix_tokens is None else [int(v) for v in text_prefix_tokens.flatten()]
    gens = []
    for p, t in grid:
        toks, stop_pos = sample_tokens(model, embeds, entry_length, stop, mode=1, top_p=p, top_k=int(top_k), temperature=t, generator=generator,
                                       head_tokens=text_prefix_tokens, desired_sentence_length=desired_sentence_length,
                                       sentence_length_factor=sentence_length_factor)
        toks, stop_pos = toks.cpu(), stop_pos.cpu()
        gens.extend(tokenizer.decode(head + toks[r, :int(stop_pos[r])].tolist()) for r in range(toks.shape[0]))
    return gens
