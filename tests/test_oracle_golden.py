"""Pins the CPU oracle against outputs of the reference itself (tests/golden/*, made by oracle/gen_golden.py).

The reference has no tests / golden vectors of its own; these are the §8c fixtures.
Tolerance: fp32, <=1e-5 abs on O(1) values (2e-5 on summed gradients).
"""
import numpy as np
import torch

from oracle import clipcap_oracle as O
from tests.util import load_golden, sd_of


def _close(a, b, tol):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    err = np.abs(a - b).max()
    assert err <= tol, f"max abs err {err} > {tol}"


def _mapper_case(name):
    g = load_golden(name)
    E, D, P, L, H, N, B = [int(v) for v in g["dims"]]
    sd = {k: v.requires_grad_(True) for k, v in sd_of(g).items()}
    x = torch.from_numpy(g["in.x"])
    out, layers, atts = O.mapper_forward(sd, x, projection_length=P, num_heads=H, num_layers=N, return_all=True)
    _close(out, g["out"], 1e-5)
    for i, a in enumerate(atts):
        _close(a, g[f"att.{i}"], 1e-6)
    loss = out.square().mean()
    loss.backward()
    _close(loss, g["loss"], 1e-5)
    for k, v in sd.items():
        _close(v.grad, g["grad." + k], 2e-5)


def test_mapper_tiny():
    _mapper_case("mapper_tiny")


def test_mapper_shape_faithful_hd96_s20():
    _mapper_case("mapper_faithful")


def test_mapper_windowed():
    g = load_golden("mapper_windowed")
    E, D, P, L, H, N, B, W = [int(v) for v in g["dims"]]
    out = O.mapper_forward(sd_of(g), torch.from_numpy(g["in.x"]), projection_length=P, num_heads=H,
                           num_layers=N, window=W)
    _close(out, g["out"], 1e-5)


def test_gpt2_tiny_logits_grads_and_mask_noop():
    g = load_golden("gpt2_tiny")
    D, n_layer, n_head, V, npos = [int(v) for v in g["cfg"]]
    sd = sd_of(g)
    sd = {k: v.requires_grad_(True) for k, v in sd.items() if k != "lm_head.weight"}
    x = torch.from_numpy(g["in.x"]).requires_grad_(True)
    logits = O.gpt2_logits(sd, x, n_head, n_layer)
    _close(logits, g["logits"], 2e-5)
    logits.square().mean().backward()
    _close(x.grad, g["grad.in.x"], 2e-5)
    for k, v in sd.items():
        if ("grad." + k) in g:
            _close(v.grad, g["grad." + k], 5e-5)
    # right-padding attention_mask has zero effect on non-pad rows (pure causal kernel is exact there)
    mask = g["mask"]
    lm = g["logits_masked"]
    ours = logits.detach().numpy()
    assert np.abs(ours[mask] - lm[mask]).max() <= 2e-5


def _train_case(mode):
    g = load_golden(f"train_{mode}")
    E, D, P, L, H, N, n_head, n_layer, V, npos = [int(v) for v in g["cfg"]]
    cfg = dict(projection_length=P, prefix_length=L, heads=H, layers=N, n_head=n_head, n_layer=n_layer)
    sd = sd_of(g)
    sd.pop("language_model.lm_head.weight", None)
    trainable = [k for k in sd if mode == "full" or k.startswith("transformer_mapper.")]
    for k in trainable:
        sd[k].requires_grad_(True)
    tokens = torch.from_numpy(g["in.tokens"])
    embeds = torch.from_numpy(g["in.embeds"])
    with torch.no_grad():
        lg = O.clipcap_logits(sd, torch.where(tokens < 0, 0, tokens), embeds, cfg=cfg)
    # rows at pad positions see the reference's attention_mask (they are never read by the loss): compare the rest
    keep = np.concatenate([np.ones((tokens.shape[0], L), bool), g["in.tokens"] >= 0], axis=1)
    assert np.abs(lg.numpy()[keep] - g["logits0"][keep]).max() <= 3e-5
    m = {k: torch.zeros_like(sd[k]) for k in trainable}
    v = {k: torch.zeros_like(sd[k]) for k in trainable}
    for step in range(3):
        for k in trainable:
            sd[k].grad = None
        loss = O.clipcap_loss(sd, tokens, embeds, cfg=cfg)
        loss.backward()
        assert abs(float(loss.detach()) - g["losses"][step]) <= 2e-5
        if step == 0:
            for k in trainable:
                if ("grad0." + k) in g:
                    _close(sd[k].grad, g["grad0." + k], 2e-5)
        lr = 1e-3 * O.linear_schedule_factor(step, 2, 6)
        assert abs(lr - g["lrs"][step]) <= 1e-12
        with torch.no_grad():
            for k in trainable:
                pn, m[k], v[k] = O.adamw_step(sd[k], sd[k].grad, m[k], v[k], step + 1, lr)
                sd[k].copy_(pn)
        for k in trainable:
            key = f"sd_after{step + 1}." + k
            if key in g:
                _close(sd[k], g[key], 5e-6)  # Adam amplifies fp32 round-off of near-zero grads (update = lr*m/sqrt(v))


def test_training_step_prefix_only_three_steps():
    _train_case("prefix_only")


def test_training_step_full_finetune_three_steps():
    _train_case("full")


def test_generate_beam_tokens_match_reference():
    g = load_golden("beam_tiny")
    D, n_layer, n_head, V, npos = [int(v) for v in g["cfg"]]
    sd = {"language_model." + k: v for k, v in sd_of(g).items()}
    cases = [str(int(c)) for c in g["cases"]] + ["T"]
    n_early = 0
    for c in cases:
        eos, entry, beam = [int(v) for v in g[f"beam{c}.meta"]]
        temp = 0.7 if c == "T" else 1.0
        toks, scores, lens, order = O.generate_beam_tokens(
            sd, torch.from_numpy(g[f"beam{c}.prefix"]), n_head=n_head, n_layer=n_layer, beam_size=beam,
            entry_length=entry, temperature=temp, stop_token=eos)
        best = toks[order[0]][: int(lens[order[0]])].numpy()
        assert np.array_equal(best, g[f"beam{c}.best"]), (c, best, g[f"beam{c}.best"])
        n_early += int(toks.shape[1] < entry)
    assert n_early >= 1, "fixture should include a beam set that stops on EOS early"


def test_filters_and_nucleus_distribution():
    g = load_golden("filters")
    lg = torch.from_numpy(g["in.logits"])
    toks = torch.from_numpy(g["in.tokens"])
    assert np.array_equal(O.top_k_top_p_filtering(lg, top_k=5).numpy(), g["topk5"])
    assert np.array_equal(O.top_k_top_p_filtering(lg, top_p=0.8).numpy(), g["topp08"])
    assert np.array_equal(O.top_k_top_p_filtering(lg, top_k=10, top_p=0.5).numpy(), g["topk10_topp05"])
    _close(O.repetition_penalty_apply(lg, toks, 1.2), g["rep12"], 1e-6)
    _close(O.sentence_length_penalty_apply(torch.from_numpy(g["in.logits2"]), toks, 11, 4, 50, 1.0), g["lenpen"], 1e-6)
    b = load_golden("beam_tiny")
    D, n_layer, n_head, V, npos = [int(v) for v in b["cfg"]]
    sd = sd_of(b)
    logits = O.gpt2_logits(sd, torch.from_numpy(g["nucleus.prefix"]), n_head, n_layer)[:, -1, :]
    _close(O.nucleus_final_p(logits, top_p=0.8), g["nucleus.final_p"], 1e-6)
