"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) — fp32 torch-CPU restatement of the ClipCap hot path.

Parity status: the reference ships NO tests or golden vectors, so this oracle is pinned
against outputs of the reference itself, imported in the build container by ``oracle/gen_golden.py``
(fixtures under ``tests/golden/``, checked by ``tests/test_oracle_golden.py``).

Every function cites the reference lines it restates (paths relative to /root/reference, or to the
installed ``transformers`` 5.15.0 ``models/gpt2/modeling_gpt2.py`` = "hf:").  State-dict key names are the
reference's so a reference checkpoint can be fed to these functions unchanged.

``rb=True`` switches on "bf16 rounding points": every tensor the HIP kernels hold in bf16 (GEMM operands,
stored qkv / attention output / MLP hidden / logits-for-backward) is rounded to bf16 at the same place the
kernels round it, with fp32 accumulation everywhere.  ``rb="fp16"`` does the same with IEEE half, the operand
type of the kernels' fp16 build (the reference's ``--fp-precision 16``).  That is the like-for-like yardstick
for the 1e-3 logits target (SURVEY.md §7 "Tolerance").  ``rb="bf16x3"`` models the split-operand build (the reference's
default ``--fp-precision 32``): a GEMM operand is worth bf16(x) + bf16(x - bf16(x)); the build keeps the activations between
kernels in fp32 there, so the other rounding points are (nearly) exact — the difference from rb=False is ~1e-5.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def _r(t: Tensor, rb) -> Tensor:
    """16-bit rounding point: identity unless rb; rb = True / "bf16" rounds to bf16, rb = "fp16" to IEEE half (the operand type of
    the kernels' --fp-precision 16 build).  Works for fp32 and fp64 tensors (the noise-floor runs evaluate the same points in fp64)."""
    if not rb:
        return t
    if rb == "bf16x3":      # split operand of the kernels' bf16x3 build: hi = bf16(x), lo = bf16(x - hi); the operand is worth hi + lo
        hi = t.to(torch.bfloat16).to(t.dtype)
        return hi + (t - hi).to(torch.bfloat16).to(t.dtype)
    if rb == "fp16x2":      # split fp16 operand: hi = fp16(x), lo = fp16(x - hi)
        hi = t.to(torch.float16).to(t.dtype)
        return hi + (t - hi).to(torch.float16).to(t.dtype)
    return t.to(torch.float16 if rb == "fp16" else torch.bfloat16).to(t.dtype)


def _m(rb, site: str):
    """Per-site rounding mode (tools/diag/site_error_budget.py): rb may be a dict {site class: mode, "default": mode} with the site
    classes "mapper", "c_attn", "attn" (stored qkv and the probabilities fed to P V), "attn.c_proj", "c_fc", "mlp.c_proj", "lm_head";
    any other rb applies to every site."""
    if isinstance(rb, dict):
        return rb.get(site, rb.get("default", False))
    return rb


def _linear(x: Tensor, w: Tensor, b: Optional[Tensor], rb: bool) -> Tensor:
    """y = x @ w.T + b with bf16-rounded operands / fp32 accumulate when rb (torch.nn.Linear layout)."""
    y = _r(x, rb) @ _r(w, rb).t()
    return y if b is None else y + b


# --------------------------------------------------------------------------------------------------
# Mapper  (clipcap/model/mapper.py:8-130, clipcap/model/attention.py:4-43)
# --------------------------------------------------------------------------------------------------

def mha_forward(p: Dict[str, Tensor], pre: str, x: Tensor, num_heads: int, rb: bool = False):
    """MultiHeadAttention.forward, self-attention, mask=None (attention.py:17-43).

    Q = x Wq^T (no bias), KV = x Wkv^T (no bias) split (b,m,2,h,d) (attention.py:24-30);
    scores einsum('bnhd,bmhd->bnmh') * hd^-0.5 (:32); softmax over keys dim=2 (:38);
    out einsum('bnmh,bmhd->bnhd') (:40); project with bias (:41).  Returns (out, attention(b,n,m,h)).
    """
    b, n, c = x.shape
    hd = c // num_heads
    q = _r(_linear(x, p[pre + "to_queries.weight"], None, rb), rb).reshape(b, n, num_heads, hd)
    kv = _r(_linear(x, p[pre + "to_keys_values.weight"], None, rb), rb).reshape(b, n, 2, num_heads, hd)
    k, v = kv[:, :, 0], kv[:, :, 1]
    att = torch.einsum("bnhd,bmhd->bnmh", q, k) * (hd ** -0.5)
    att = att.softmax(dim=2)
    out = torch.einsum("bnmh,bmhd->bnhd", _r(att, rb), v).reshape(b, n, c)   # the MFMA kernel feeds P to the PV product in bf16
    out = _r(out, rb)
    out = _linear(out, p[pre + "project.weight"], p[pre + "project.bias"], rb)
    return out, att


def transformer_layer_forward(p: Dict[str, Tensor], pre: str, x: Tensor, num_heads: int, rb: bool = False):
    """TransformerLayer.forward (mapper.py:107-110): x += attn(LN1 x); x += fc2(relu(fc1(LN2 x))).

    LayerNorm eps 1e-5 affine (torch default; mapper.py:96,98); MLP hidden = int(D*2.0) because
    Transformer passes mlp_ratio=2. (mapper.py:10,39); dropout p=0 (identity).
    """
    d = x.shape[-1]
    h = F.layer_norm(x, (d,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], 1e-5)
    a, att = mha_forward(p, pre + "attn.", h, num_heads, rb)
    x = x + a
    h = F.layer_norm(x, (d,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], 1e-5)
    h = torch.relu(_linear(h, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"], rb))
    h = _r(h, rb)
    x = x + _linear(h, p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"], rb)
    return x, att


def mapper_forward(p: Dict[str, Tensor], x: Tensor, *, projection_length: int, num_heads: int,
                   num_layers: int, pre: str = "", window: int = 1, rb: bool = False,
                   return_all: bool = False):
    """TransformerMapper.forward (mapper.py:122-130) / TransformerMapperWindowed.forward (:148-160).

    x (B,E) [window==1] or (B,W,E): linear -> view (B, W*P, D) (+ pos_embeddings if present) ->
    cat learned prefix_const (L,D) -> N layers -> rows [W*P:].
    """
    bsz = x.shape[0]
    rb = _m(rb, "mapper")
    wgt, bias = p[pre + "linear.weight"], p[pre + "linear.bias"]
    proj = _linear(x, wgt, bias, rb).reshape(bsz, window * projection_length, -1)
    if (pre + "pos_embeddings") in p:
        proj = proj + p[pre + "pos_embeddings"].unsqueeze(0)
    const = p[pre + "prefix_const"]
    h = torch.cat((proj, const.unsqueeze(0).expand(bsz, *const.shape)), dim=1)
    layers, atts = [h], []
    for i in range(num_layers):
        h, att = transformer_layer_forward(p, f"{pre}transformer.layers.{i}.", h, num_heads, rb)
        layers.append(h)
        atts.append(att)
    out = h[:, window * projection_length:]
    if return_all:
        return out, layers, atts
    return out


# --------------------------------------------------------------------------------------------------
# GPT-2  (hf: modeling_gpt2.py:54-72 attention, :144-226 GPT2Attention, :229-243 MLP, :262-310 block,
#         :514-634 GPT2Model.forward, :650-725 LM head; activation gelu_new; tied lm_head)
# --------------------------------------------------------------------------------------------------

def gelu_new(x: Tensor) -> Tensor:
    """transformers.activations.NewGELUActivation: 0.5x(1+tanh(sqrt(2/pi)(x+0.044715x^3)))."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def _conv1d(x: Tensor, w: Tensor, b: Tensor, rb: bool) -> Tensor:
    """HF Conv1D: y = x @ W + b with W (in,out) (pytorch_utils.Conv1D.forward)."""
    return _r(x, rb) @ _r(w, rb) + b


def gpt2_block_forward(p, pre, x, n_head, rb=False, past_kv=None, drop=None, layer=0):
    """GPT2Block.forward (hf :262-310): pre-LN; causal MHA with scale hd^-0.5 (:97-98, :54-72); MLP gelu_new.

    past_kv: optional (K,V) each (B,H,ctx,hd) for the KV-cached decode restatement; returns new (K,V).
    drop: train-mode dropout with EXTERNALLY supplied keep masks (hf attn_pdrop on the attention probabilities inside sdpa,
    resid_pdrop after both c_proj, :214-226, :241): {"p_attn", "p_resid", "attn": [mask (B,H,T,T)], "resid_attn": [(B,T,D)],
    "resid_mlp": [(B,T,D)]}; kept entries are scaled by 1/(1-p) like torch.nn.functional.dropout.
    """
    bsz, t, d = x.shape
    hd = d // n_head
    h = F.layer_norm(x, (d,), p[pre + "ln_1.weight"], p[pre + "ln_1.bias"], 1e-5)
    qkv = _r(_conv1d(h, p[pre + "attn.c_attn.weight"], p[pre + "attn.c_attn.bias"], _m(rb, "c_attn")), _m(rb, "attn"))
    q, k, v = qkv.split(d, dim=2)
    q = q.reshape(bsz, t, n_head, hd).transpose(1, 2)
    k = k.reshape(bsz, t, n_head, hd).transpose(1, 2)
    v = v.reshape(bsz, t, n_head, hd).transpose(1, 2)
    if past_kv is not None:
        k = torch.cat((past_kv[0], k), dim=2)
        v = torch.cat((past_kv[1], v), dim=2)
    ctx = k.shape[2]
    att = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
    # causal: query at absolute position (ctx - t + i) sees keys <= that position
    qi = torch.arange(ctx - t, ctx).unsqueeze(1)
    kj = torch.arange(ctx).unsqueeze(0)
    att = att.masked_fill(kj > qi, float("-inf"))
    att = att.softmax(dim=-1)
    if drop is not None and drop.get("p_attn", 0.0) > 0:
        att = att * drop["attn"][layer] / (1.0 - drop["p_attn"])
    a = _r((_r(att, _m(rb, "attn")) @ v).transpose(1, 2).reshape(bsz, t, d), _m(rb, "attn.c_proj"))
    y = _conv1d(a, p[pre + "attn.c_proj.weight"], p[pre + "attn.c_proj.bias"], _m(rb, "attn.c_proj"))
    if drop is not None and drop.get("p_resid", 0.0) > 0:
        y = y * drop["resid_attn"][layer] / (1.0 - drop["p_resid"])
    x = x + y
    h = F.layer_norm(x, (d,), p[pre + "ln_2.weight"], p[pre + "ln_2.bias"], 1e-5)
    h = _r(gelu_new(_conv1d(h, p[pre + "mlp.c_fc.weight"], p[pre + "mlp.c_fc.bias"], _m(rb, "c_fc"))), _m(rb, "mlp.c_proj"))
    y = _conv1d(h, p[pre + "mlp.c_proj.weight"], p[pre + "mlp.c_proj.bias"], _m(rb, "mlp.c_proj"))
    if drop is not None and drop.get("p_resid", 0.0) > 0:
        y = y * drop["resid_mlp"][layer] / (1.0 - drop["p_resid"])
    x = x + y
    return x, (k, v)


def gpt2_hidden(p, inputs_embeds, n_head, n_layer, pre="transformer.", rb=False, past=None, pos_offset=0, drop=None):
    """GPT2Model.forward with inputs_embeds (hf :514-634): + wpe[arange(T)+past] (:571-577), blocks, ln_f.

    Right-padding attention_mask is not restated: it has exactly zero effect on non-pad rows under a
    causal mask (pinned by tests/golden gpt2 'masked' fixture).
    """
    t = inputs_embeds.shape[1]
    x = inputs_embeds + p[pre + "wpe.weight"][pos_offset:pos_offset + t].unsqueeze(0)
    if drop is not None and drop.get("p_embd", 0.0) > 0:          # hf :586 self.drop(hidden_states)
        x = x * drop["embd"] / (1.0 - drop["p_embd"])
    new_past = []
    for i in range(n_layer):
        x, kv = gpt2_block_forward(p, f"{pre}h.{i}.", x, n_head, rb, None if past is None else past[i], drop, i)
        new_past.append(kv)
    d = x.shape[-1]
    x = F.layer_norm(x, (d,), p[pre + "ln_f.weight"], p[pre + "ln_f.bias"], 1e-5)
    return x, new_past


def gpt2_logits(p, inputs_embeds, n_head, n_layer, pre="", rb=False, drop=None):
    """GPT2LMHeadModel.forward (hf :650-725): logits = hidden @ wte^T (tied, :638, :703)."""
    h, _ = gpt2_hidden(p, inputs_embeds, n_head, n_layer, pre + "transformer.", rb, drop=drop)
    return _r(h, _m(rb, "lm_head")) @ _r(p[pre + "transformer.wte.weight"], _m(rb, "lm_head")).t()


# --------------------------------------------------------------------------------------------------
# ClipCapModel  (clipcap/model/model.py:43-58 forward, :94-113 training_step)
# --------------------------------------------------------------------------------------------------

def clipcap_logits(p, tokens, embeds, *, cfg, rb=False, drop=None):
    """ClipCapModel.forward (model.py:43-58): wte(tokens); mapper; cat [prefix; tok]; GPT-2 -> logits (B,T,V)."""
    wte = p["language_model.transformer.wte.weight"]
    tok_emb = wte[tokens]
    prefix = mapper_forward(p, embeds, projection_length=cfg["projection_length"], num_heads=cfg["heads"],
                            num_layers=cfg["layers"], pre="transformer_mapper.", window=cfg.get("window", 1), rb=rb)
    x = torch.cat((prefix, tok_emb), dim=1)
    return gpt2_logits(p, x, cfg["n_head"], cfg["n_layer"], pre="language_model.", rb=rb, drop=drop)


def clipcap_loss(p, tokens_padded, embeds, *, cfg, rb=False, denom: Optional[float] = None, drop=None):
    """ClipCapModel.training_step (model.py:94-113).

    mask = tokens>=0; pads -> 0 (:103-104); logits[:, L-1:-1] (:108); cross_entropy(ignore_index=0) mean over
    non-ignored targets (:109) — token id 0 is ignored too.  ``denom`` overrides the divisor (global
    non-ignored count for the N-rank == 1-rank DDP spec).
    """
    tokens = tokens_padded.clone()
    tokens[tokens < 0] = 0
    logits = clipcap_logits(p, tokens, embeds, cfg=cfg, rb=rb, drop=drop)
    L = cfg["prefix_length"]
    lg = logits[:, L - 1:-1]
    if denom is None:
        return F.cross_entropy(lg.reshape(-1, lg.shape[-1]), tokens.flatten(), ignore_index=0)
    s = F.cross_entropy(lg.reshape(-1, lg.shape[-1]), tokens.flatten(), ignore_index=0, reduction="sum")
    return s / denom


# --------------------------------------------------------------------------------------------------
# Optimizer + schedule  (clipcap/model/model.py:67-91; torch.optim.AdamW defaults; HF linear schedule)
# --------------------------------------------------------------------------------------------------

def linear_schedule_factor(step: int, warmup: int, total: int) -> float:
    """transformers.optimization._get_linear_schedule_with_warmup_lr_lambda."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    return max(0.0, float(total - step) / float(max(1, total - warmup)))


def adamw_step(param, grad, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.01):
    """torch.optim.AdamW single-tensor math (decoupled decay; bias correction; eps outside sqrt-bias-corr)."""
    param = param * (1.0 - lr * wd)
    m = beta1 * m + (1 - beta1) * grad
    v = beta2 * v + (1 - beta2) * grad * grad
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    param = param - (lr / bc1) * (m / denom)
    return param, m, v


# --------------------------------------------------------------------------------------------------
# Decode  (clipcap/inference/base.py:55-132 generate_beam; :9-52 filters)
# --------------------------------------------------------------------------------------------------

def beam_update(logits, scores, seq_lengths, has_stopped, *, beam_size, temperature=1.0, stop_token=50256):
    """One beam update of generate_beam (inference/base.py:82-119) for one sample.  logits (1, V) on the first step (scores is
    None), (beam, V) afterwards.  Returns (next_tokens (beam,), src (beam,) or None, scores, seq_lengths, has_stopped)."""
    logits = logits / (temperature if temperature > 0 else 1.0)
    logits = logits.softmax(-1).log()
    if scores is None:
        scores, next_tokens = logits.topk(beam_size, -1)                      # :86-87
        next_tokens, scores = next_tokens.permute(1, 0).squeeze(1), scores.squeeze(0)
        src = None
    else:
        logits[has_stopped] = -float("inf")                                   # :96-97
        logits[has_stopped, 0] = 0
        scores_sum = scores[:, None] + logits
        seq_lengths = seq_lengths.clone()
        seq_lengths[~has_stopped] += 1                                        # :99
        avg = scores_sum / seq_lengths[:, None]
        avg, next_tokens = avg.view(-1).topk(beam_size, -1)
        src = torch.div(next_tokens, scores_sum.shape[1], rounding_mode="trunc")
        seq_lengths = seq_lengths[src]
        next_tokens = next_tokens % scores_sum.shape[1]
        scores = avg * seq_lengths                                            # :110
        has_stopped = has_stopped[src]
    has_stopped = has_stopped + next_tokens.eq(stop_token)                    # :119
    return next_tokens, src, scores, seq_lengths, has_stopped


def generate_beam_tokens(p, embeds, *, n_head, n_layer, beam_size=5, entry_length=67, temperature=1.0,
                         stop_token=50256, pre="language_model.", rb=False, trace: Optional[list] = None, kv_cache: bool = False, rounds: int = 1):
    """Token-level restatement of generate_beam (inference/base.py:55-132) for one sample.

    embeds (1, L, D).  Full re-forward per step like the reference (:81) — the KV-cached product path must
    reproduce these tokens.  ``kv_cache=True`` is the same search with the keys / values of earlier positions kept
    (the arithmetic of every step is unchanged; used as the "cached" CPU baseline of bench.py and to cross-check
    the cache logic).  Returns (tokens (beam, n) int64, scores (beam,), seq_lengths (beam,), order);
    the reference returns tokenizer.decode(tokens[order[0]][:int(seq_lengths[order[0]])]) (:123-130).
    ``rounds`` > 1: the reference's number_to_generate loop (:79-130) — the entry_length loop re-entered with live state, ``scores``
    rebound to scores / seq_lengths at the end of every round (:123); a list of those 4-tuples, one per round, is returned instead.
    """
    wte = p[pre + "transformer.wte.weight"]
    tokens = None
    scores = None
    seq_lengths = torch.ones(beam_size)
    has_stopped = torch.zeros(beam_size, dtype=torch.bool)
    past, pos, new = None, 0, embeds
    results = []
    for _round in range(max(1, rounds)):
        for _ in range(entry_length):
            if kv_cache:
                h, past = gpt2_hidden(p, new, n_head, n_layer, pre + "transformer.", rb, past=past, pos_offset=pos)
                pos += new.shape[1]
                logits = (_r(h[:, -1, :], rb) @ _r(wte, rb).t())
            else:
                logits = gpt2_logits(p, embeds, n_head, n_layer, pre=pre, rb=rb)[:, -1, :]
            next_tokens, src, scores, seq_lengths, has_stopped = beam_update(logits, scores, seq_lengths, has_stopped, beam_size=beam_size,
                                                                             temperature=temperature, stop_token=stop_token)
            if src is None:
                embeds = embeds.expand(beam_size, *embeds.shape[1:])
                tokens = next_tokens.unsqueeze(1)
                if kv_cache:
                    past = [(k.expand(beam_size, *k.shape[1:]), v.expand(beam_size, *v.shape[1:])) for k, v in past]
            else:
                tokens = torch.cat((tokens[src], next_tokens.unsqueeze(1)), dim=1)
                embeds = embeds[src]
                if kv_cache:
                    past = [(k[src], v[src]) for k, v in past]
            nxt = wte[next_tokens].view(embeds.shape[0], 1, -1)
            new = nxt
            if not kv_cache:
                embeds = torch.cat((embeds, nxt), dim=1)
            if trace is not None:
                trace.append(dict(tokens=tokens.clone(), scores=scores.clone(), seq_lengths=seq_lengths.clone(),
                                  has_stopped=has_stopped.clone()))
            if has_stopped.all():
                break
        scores = scores / seq_lengths                           # :123 — the NEXT round's running scores are these
        order = scores.argsort(descending=True)
        results.append((tokens.clone(), scores.clone(), seq_lengths.clone(), order))
    return results[0] if rounds <= 1 else results


def top_k_top_p_filtering(logits: Tensor, top_k: int = 0, top_p: float = 0.0, filter_value=-float("inf")) -> Tensor:
    """inference/base.py:9-38 (1-D logits; returns a filtered copy — the reference filters in place)."""
    logits = logits.clone()
    top_k = min(top_k, logits.size(-1))
    if top_k > 0:
        kth = torch.topk(logits, top_k)[0][..., -1, None]
        logits[logits < kth] = filter_value
    if top_p > 0.0:
        sorted_logits, sorted_idx = torch.sort(logits, descending=True)
        cum = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
        remove = cum > top_p
        remove[..., 1:] = remove[..., :-1].clone()
        remove[..., 0] = False
        logits[sorted_idx[remove]] = filter_value
    return logits


def repetition_penalty_apply(logits: Tensor, tokens: Tensor, penalty: float) -> Tensor:
    """inference/base.py:40-44."""
    logits = logits.clone()
    t = torch.gather(logits, -1, tokens)
    t = torch.where(t < 0, t * penalty, t / penalty)
    logits.scatter_(-1, tokens, t)
    return logits


def sentence_length_penalty_apply(logits, tokens, stop_token, current_length, desired_length, length_factor):
    """inference/base.py:46-55 (note: compares the gathered *logit value* with stop_token, as the reference does)."""
    logits = logits.clone()
    penalty = (current_length / desired_length) * length_factor
    t = torch.gather(logits, -1, tokens)
    t = torch.where(t == stop_token, t * penalty, t)
    logits.scatter_(-1, tokens, t)
    return logits


def nucleus_final_p(logits: Tensor, top_p: float = 0.8, top_k: Optional[int] = None) -> Tensor:
    """Pre-sampling distribution of generate_nucleus_sampling (inference/base.py:165-181), logits (n,V)."""
    if top_k is None:
        top_k = logits.shape[-1]
    if top_p is None:
        top_p = 1.0
    pr, idx_sorted = F.softmax(logits, dim=-1).topk(top_k, dim=-1)
    cum = pr.cumsum(dim=-1)
    thr = top_p + torch.zeros((len(pr), 1))
    idx = torch.searchsorted(cum, thr).clip(max=top_k - 1).squeeze()
    cut = cum[torch.arange(len(cum)), idx]
    cens = (cum <= cut[:, None]) * pr
    ren = cens / cens.sum(dim=-1, keepdims=True)
    final = torch.zeros_like(logits)
    rows = torch.arange(len(pr)).unsqueeze(1).repeat(1, top_k)
    final[rows, idx_sorted] = ren.to(final.dtype)
    return final


def no_beam_step_distribution(logits: Tensor, history: Optional[Tensor], *, top_p: float, top_k, temperature: float,
                              repetition_penalty: float, stop_token: Optional[int] = None, desired_sentence_length: int = 50,
                              sentence_length_factor: float = 1.0) -> Tensor:
    """Pre-sampling distribution of one step of inference/no_beam.py:35-63 for 1-D ``logits`` (the variant generate() uses):
    repetition penalty over ``history`` = text_prefix_tokens ++ generated (:45-48) -> / temperature (:51) ->
    top_k_top_p_filtering (:52) -> sentence-length penalty (:55-60, value comparison as in the reference) -> softmax (:63)."""
    lg = logits.clone()
    has_hist = history is not None and history.numel() > 0
    if repetition_penalty != 1.0 and has_hist:
        lg = repetition_penalty_apply(lg, history, repetition_penalty)
    lg = lg / (temperature if temperature > 0 else 1.0)
    lg = top_k_top_p_filtering(lg, top_k=int(top_k), top_p=top_p)
    if has_hist and stop_token is not None:
        lg = sentence_length_penalty_apply(lg, history, stop_token, history.numel(), desired_sentence_length, sentence_length_factor)
    return F.softmax(lg, dim=-1)


def sampling_trace(p, embeds, forced: List[int], *, n_head, n_layer, rule: str, head: Optional[Tensor] = None, pre="language_model.",
                   rb=False, **kw) -> List[Tensor]:
    """Per-step pre-sampling distributions of the reference's sampling loops when the drawn tokens are ``forced`` (full re-forward
    per step like no_beam.py:38 / nucleus_sampling.py:35).  rule "no_beam": no_beam.py:35-75 (kw: top_p, top_k, temperature,
    repetition_penalty, stop_token); rule "nucleus": nucleus_sampling.py:34-56 (kw: top_p, top_k (0 = all), temperature).
    embeds (1, L, D) already holds the text prefix's embeddings; ``head`` (H0,) are the text_prefix_tokens of the penalty history."""
    wte = p[pre + "transformer.wte.weight"]
    hist = head.clone() if head is not None and head.numel() else None
    out = []
    for tok in forced:
        logits = gpt2_logits(p, embeds, n_head, n_layer, pre=pre, rb=rb)[0, -1, :]
        if rule == "no_beam":
            out.append(no_beam_step_distribution(logits, hist, **kw))
        else:
            t = kw.get("temperature", 1.0)
            out.append(nucleus_final_p((logits / (t if t > 0 else 1.0)).unsqueeze(0), top_p=kw.get("top_p", 0.8),
                                       top_k=(kw.get("top_k") or None))[0])
        nt = torch.tensor([tok])
        hist = nt if hist is None else torch.cat((hist, nt))
        embeds = torch.cat((embeds, wte[nt].unsqueeze(0)), dim=1)
    return out
