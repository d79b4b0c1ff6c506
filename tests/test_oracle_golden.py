"""Pins the CPU oracle against outputs of the reference itself (tests/golden/*, made by oracle/gen_golden.py).

The reference has no tests / golden vectors of its own; these are the §8c fixtures.
Tolerance: fp32, <=1e-5 abs on O(1) values (2e-5 on summed gradients).
"""
import numpy as np
import pytest
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
    _mapper_case("mapper_hd96")


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


# ---------------------------------------------------------------------------------------------------------------------------
# round-2 fixtures: shape-faithful cases with seeded parameters (tests/seeded.py), sampling variants, medium-width beam search
# ---------------------------------------------------------------------------------------------------------------------------

def test_mapper_shape_faithful_config2_one_layer():
    """E=512, D=768, P=L=10, H=8, N=1, B=2 — output, attention probabilities and every gradient of the reference."""
    from tests import seeded
    from tests.util import sampled
    g = load_golden("mapper_faithful")
    E, D, P, L, H, N, B, seed = [int(v) for v in g["dims"]]
    msd = seeded.state_dict(seeded.mapper_shapes(E, D, P, L, N), seed)
    assert np.array_equal(seeded.checksum(msd), g["param_checksum"])
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in msd.items()}
    out, _, atts = O.mapper_forward(sd, torch.from_numpy(g["in.x"]), projection_length=P, num_heads=H, num_layers=N, return_all=True)
    _close(out, g["out"], 2e-5)
    _close(atts[0], g["att.0"], 1e-5)
    out.square().mean().backward()
    for k, v in sd.items():
        nrm, smp = sampled(v.grad)
        assert abs(nrm - float(g[f"grad.{k}.norm"])) <= 1e-4 * float(g[f"grad.{k}.norm"]) + 1e-9, k
        assert np.abs(smp - g[f"grad.{k}.sample"]).max() <= 2e-5 * max(1.0, np.abs(g[f"grad.{k}.sample"]).max()), k


@pytest.mark.parametrize("name", ["config2_full", "config3_full", "config4_full"])
def test_full_depth_model_logits_loss_grads(name):
    """BASELINE configs[1] / configs[2] / configs[3] architectures at full depth (8-layer mapper + 12-layer GPT-2-small, frozen and
    full finetune; E=1024 mapper + 24-layer GPT-2-medium), B=2: the oracle against the reference's logits, prefix, loss and gradients."""
    from tests.util import sampled, seeded_full_model
    from tests.seeded import sample_idx
    g = load_golden(name)
    sd, cfg, dims = seeded_full_model(g)
    train = [k for k in sd if dims["full"] or k.startswith("transformer_mapper.")]
    for k in train:
        sd[k].requires_grad_(True)
    tokens, embeds = torch.from_numpy(g["in.tokens"]), torch.from_numpy(g["in.embeds"])
    L, V, cap = dims["L"], dims["V"], tokens.shape[1]
    with torch.no_grad():
        logits = O.clipcap_logits(sd, tokens.clamp_min(0), embeds, cfg=cfg)
    # the reference was run with its right-padding attention mask: rows of pad positions differ by construction (and are never
    # read by the loss); every other row is mask-independent under the causal mask
    valid = torch.cat((torch.ones(tokens.shape[0], L, dtype=torch.bool), tokens.ge(0)), dim=1)
    _close(logits[:, :, sample_idx(V, 1024)] * valid[:, :, None], g["logits.cols"] * valid[:, :, None].numpy(), 3e-5)
    rows = [L - 1, L + 7, L + cap - 2]
    _close(logits[:, rows, :] * valid[:, rows, None], g["logits.rows"] * valid[:, rows, None].numpy(), 3e-5)
    loss = O.clipcap_loss(sd, tokens, embeds, cfg=cfg)
    assert abs(float(loss) - float(g["loss"])) <= 2e-5
    loss.backward()
    n = 0
    for k in train:
        key = "grad0." + k
        if key + ".norm" not in g:
            continue
        nrm, smp = sampled(sd[k].grad)
        ref = float(g[key + ".norm"])
        assert abs(nrm - ref) <= 2e-4 * ref + 1e-10, (k, nrm, ref)
        assert np.abs(smp - g[key + ".sample"]).max() <= 5e-5 * max(np.abs(g[key + ".sample"]).max(), 1e-8) + 1e-9, k
        n += 1
    assert n >= (100 if dims["full"] else 90)


def test_sampling_variants_step_distributions():
    """inference/no_beam.py (repetition penalty, '.' stop, text prefix in the history) and inference/nucleus_sampling.py: the
    pre-sampling distribution of EVERY step of the reference's loop (torch.multinomial patched to a forced token sequence)."""
    g = load_golden("sampling_steps")
    b = load_golden("gpt2_tiny")
    D, n_layer, n_head, V, npos = [int(v) for v in b["cfg"]]
    sd = {"language_model." + k: v for k, v in sd_of(b).items()}
    sd["language_model.transformer.wte.weight"] = sd["language_model.transformer.wte.weight"] * float(g["wte_scale"])
    wte = sd["language_model.transformer.wte.weight"]
    for case, rule in (("no_beam", "no_beam"), ("no_beam_topk", "no_beam"), ("nucleus", "nucleus")):
        top_p, top_k, temp, pen, stop = [float(v) for v in g[case + ".kw"]]
        head = torch.from_numpy(g[case + ".head"]).reshape(-1)
        emb = torch.from_numpy(g[case + ".prefix"])
        if head.numel():
            emb = torch.cat((emb, wte[head].unsqueeze(0)), dim=1)
        kw = dict(top_p=top_p, top_k=int(top_k), temperature=temp)
        if rule == "no_beam":
            kw.update(repetition_penalty=pen, stop_token=int(stop))
        tr = O.sampling_trace(sd, emb, [int(t) for t in g[case + ".forced"]], n_head=n_head, n_layer=n_layer, rule=rule, head=head, **kw)
        got = torch.stack(tr).numpy()
        assert got.shape == g[case + ".probs"].shape
        assert np.array_equal(got > 0, g[case + ".probs"] > 0), case
        assert np.abs(got - g[case + ".probs"]).max() <= 2e-6, case
        assert np.array_equal(np.concatenate([head.numpy(), g[case + ".forced"]]), g[case + ".text"]), case


def test_sentence_length_penalty_where_it_fires():
    """no_beam.py:55-60 / utils.py:40-51: rows whose history tokens carry a filtered logit equal to float(stop id), so that the
    reference's value comparison is true (tests/golden/length_penalty.npz, oracle/gen_golden.py (13))."""
    g = load_golden("length_penalty")
    n = len([k for k in g if k.endswith(".logits")])
    assert n >= 6
    for ci in range(n):
        stop, temp, rep, top_p, top_k, want_len, factor, fired = [float(v) for v in g[f"c{ci}.kw"]]
        lg, hist = torch.from_numpy(g[f"c{ci}.logits"]), torch.from_numpy(g[f"c{ci}.hist"])
        got = O.no_beam_step_distribution(lg, hist, top_p=top_p, top_k=int(top_k), temperature=temp, repetition_penalty=rep, stop_token=int(stop),
                                          desired_sentence_length=int(want_len), sentence_length_factor=factor).numpy()
        ref = np.zeros_like(got)
        ref[g[f"c{ci}.idx"]] = g[f"c{ci}.probs"]
        assert np.array_equal(got > 0, ref > 0), ci
        assert np.abs(got - ref).max() <= 1e-7, ci
        off = O.no_beam_step_distribution(lg, hist, top_p=top_p, top_k=int(top_k), temperature=temp, repetition_penalty=rep).numpy()
        if abs(hist.numel() / want_len * factor - 1.0) > 1e-6:
            assert np.abs(off - got).max() > 1e-4, ci


def test_beam_search_medium_width():
    """BASELINE configs[4] width (D=1024, 16 heads, V=50257; 4 layers): the reference's beam-5 captions, incl. runs that stop on EOS."""
    from tests import seeded
    g = load_golden("beam_medium")
    D, NL, n_head, V, NPOS, seed = [int(v) for v in g["cfg"]]
    gsd = seeded.state_dict(seeded.gpt2_shapes(D, NL, V, NPOS), seed)
    gsd["transformer.wte.weight"] = gsd["transformer.wte.weight"] * float(g["wte_scale"])
    assert np.array_equal(seeded.checksum(gsd), g["param_checksum"])
    sd = {"language_model." + k: torch.from_numpy(v) for k, v in gsd.items()}
    torch.set_num_threads(max(1, min(8, torch.get_num_threads())))
    for case in ("beam0a", "beam0b", "beam1a", "beam1b"):
        eos, entry, beam = [int(v) for v in g[case + ".meta"]]
        toks, sc, lens, order = O.generate_beam_tokens(sd, torch.from_numpy(g[case + ".prefix"]), n_head=n_head, n_layer=NL, beam_size=beam,
                                                       entry_length=entry, stop_token=eos)
        best = toks[order[0]][: int(lens[order[0]])].numpy()
        assert np.array_equal(best, g[case + ".best"]), (case, best, g[case + ".best"])


def test_beam_search_medium_depth():
    """BASELINE configs[4] at GPT-2-medium DEPTH (24 layers, D=1024, 16 heads, V=50257): the reference's beam-5 caption of prefix 0
    (tests/golden/beam_deep.npz), without and with a stop token that freezes beams mid-way.  (Prefix 1 is covered by the GPU tests;
    two 24-layer searches keep this suite at a few minutes.)"""
    from tests import seeded
    g = load_golden("beam_deep")
    D, NL, n_head, V, NPOS, seed = [int(v) for v in g["cfg"]]
    gsd = seeded.state_dict(seeded.gpt2_shapes(D, NL, V, NPOS), seed)
    gsd["transformer.wte.weight"] = gsd["transformer.wte.weight"] * float(g["wte_scale"])
    assert np.array_equal(seeded.checksum(gsd), g["param_checksum"])
    sd = {"language_model." + k: torch.from_numpy(v) for k, v in gsd.items()}
    torch.set_num_threads(max(1, min(8, torch.get_num_threads())))
    for case in ("beam0a", "beam0b"):
        eos, entry, beam = [int(v) for v in g[case + ".meta"]]
        toks, sc, lens, order = O.generate_beam_tokens(sd, torch.from_numpy(g[case + ".prefix"]), n_head=n_head, n_layer=NL, beam_size=beam,
                                                       entry_length=entry, stop_token=eos)
        best = toks[order[0]][: int(lens[order[0]])].numpy()
        assert np.array_equal(best, g[case + ".best"]), (case, best, g[case + ".best"])


def test_beam_search_number_to_generate_rounds():
    """inference/base.py:79-130 with number_to_generate = 3 (tests/golden/beam_multi.npz): the oracle's ``rounds`` reproduces the reference's
    three consecutive generations — beams that keep growing, beams frozen mid-way, and rounds that start with every beam stopped."""
    from tests import seeded
    g = load_golden("beam_multi")
    D, NL, n_head, V, NPOS, seed = [int(v) for v in g["cfg"]]
    gsd = seeded.state_dict(seeded.gpt2_shapes(D, NL, V, NPOS), seed)
    gsd["transformer.wte.weight"] = gsd["transformer.wte.weight"] * float(g["wte_scale"])
    assert np.array_equal(seeded.checksum(gsd), g["param_checksum"])
    sd = {"language_model." + k: torch.from_numpy(v) for k, v in gsd.items()}
    torch.set_num_threads(max(1, min(8, torch.get_num_threads())))
    for case in ("multi0a", "multi0c", "multi1b", "multi1c"):
        eos, entry, beam, ng = [int(v) for v in g[case + ".meta"]]
        res = O.generate_beam_tokens(sd, torch.from_numpy(g[case + ".prefix"]), n_head=n_head, n_layer=NL, beam_size=beam, entry_length=entry,
                                     stop_token=eos, rounds=ng, kv_cache=True)
        assert len(res) == ng
        for r, (toks, sc, lens, order) in enumerate(res):
            best = toks[order[0]][: int(lens[order[0]])].numpy()
            assert np.array_equal(best, g[f"{case}.gen{r}"]), (case, r, best, g[f"{case}.gen{r}"])


def test_beam_search_varied_captions_medium_depth():
    """tests/golden/beam_varied.npz (24 layers, wpe x8, wte x0.5, temperature 1.5): the reference's non-repeating caption of prefix 0,
    without and with a stop token — beam ranking under competition at full depth (the other prefix is covered by the GPU tests)."""
    from tests import seeded
    g = load_golden("beam_varied")
    D, NL, n_head, V, NPOS, seed = [int(v) for v in g["cfg"]]
    gsd = seeded.state_dict(seeded.gpt2_shapes(D, NL, V, NPOS), seed)
    gsd["transformer.wte.weight"] = gsd["transformer.wte.weight"] * float(g["wte_scale"])
    gsd["transformer.wpe.weight"] = gsd["transformer.wpe.weight"] * float(g["wpe_scale"])
    assert np.array_equal(seeded.checksum(gsd), g["param_checksum"])
    sd = {"language_model." + k: torch.from_numpy(v) for k, v in gsd.items()}
    torch.set_num_threads(max(1, min(8, torch.get_num_threads())))
    for case in ("beam0a", "beam0b"):
        eos, entry, beam = [int(v) for v in g[case + ".meta"]]
        assert len(set(int(t) for t in g["beam0a.best"])) >= 8
        toks, sc, lens, order = O.generate_beam_tokens(sd, torch.from_numpy(g[case + ".prefix"]), n_head=n_head, n_layer=NL, beam_size=beam,
                                                       entry_length=entry, stop_token=eos, temperature=float(g["temperature"]), kv_cache=True)
        best = toks[order[0]][: int(lens[order[0]])].numpy()
        assert np.array_equal(best, g[case + ".best"]), (case, best, g[case + ".best"])
