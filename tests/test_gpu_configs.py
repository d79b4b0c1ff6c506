"""Parity of the BASELINE configurations the round-1 suite did not reach, all through the C ABI, against the pinned oracle AND
against outputs of the reference itself (tests/golden/config2_full, config4_full, beam_medium; parameters regenerated from
tests/seeded.py, oracle pinned to the same fixtures on CPU by tests/test_oracle_golden.py):

  * configs[1] at FULL depth: 8-layer mapper + 12-layer GPT-2-small, B=2, cap=40 — logits, loss, every mapper gradient;
  * configs[3]: E=1024 -> D=1024 mapper (hd 128) + 24-layer GPT-2-medium (16 heads), full finetune — logits, loss, mapper and
    GPT-2 gradients at B=2, plus size-independent properties of the B=128 step;
  * configs[4]: beam-5 KV-cached decode at GPT-2-medium size — KV cache == re-forward at 24 layers, 64 prefixes batched ==
    per-sample, token-exact captions vs the reference / the like-for-like oracle at medium width.

Tolerances: like-for-like = oracle evaluated with the kernels' bf16 rounding points (rb=True); the fp32 reference is the
looser yardstick (the reference's own bf16-autocast drift is 2.8e-2 on logits).  Integer results are exact.
"""
import numpy as np
import pytest
import torch

from oracle import clipcap_oracle as O
from tests.seeded import sample_idx
from tests.util import load_golden, sampled, seeded_full_model

pytestmark = pytest.mark.gpu


def _engines(sd, dims, precision=None):
    from clipcap_amd.engine import ClipCapEngine, Gpt2Engine, MapperEngine
    me = MapperEngine(dims["E"], dims["D"], dims["L"], dims["P"], dims["H"], dims["N"], device="cuda", precision=precision)
    ge = Gpt2Engine(dims["D"], dims["n_head"], dims["NL"], dims["V"], dims["NPOS"], device="cuda", precision=precision)
    for pre, eng in (("transformer_mapper.", me), ("language_model.", ge)):
        for k, v in eng.views(eng.arena.w32).items():
            v.copy_(sd[pre + k])
    return me, ge, ClipCapEngine(me, ge, train_lm=dims["full"])


def _rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-20))


def _full_model_case(name, tol, precision=None):
    """precision None = bf16 operands (oracle with bf16 rounding points), 16 = fp16 operands (oracle with fp16 rounding points; the
    gradients in the arenas then carry the engine's loss scale), 32 = split-bf16 operands (the reference's default precision)."""
    g = load_golden(name)
    sd, cfg, dims = seeded_full_model(g)
    me, ge, eng = _engines(sd, dims, precision)
    rb = {16: "fp16", 32: "bf16x3"}.get(precision, True)
    pts = {16: "fp16 points", 32: "split-bf16 operands"}.get(precision, "bf16 points")
    tokens, embeds = torch.from_numpy(g["in.tokens"]), torch.from_numpy(g["in.embeds"])
    L, V, cap, B = dims["L"], dims["V"], tokens.shape[1], tokens.shape[0]
    valid = torch.cat((torch.ones(B, L, dtype=torch.bool), tokens.ge(0)), dim=1)
    rows = [L - 1, L + 7, L + cap - 2]

    # ---- forward: prefix, logits of every row (ClipCapModel.forward, model.py:43-58) ----
    prefix = me.forward(embeds.cuda())
    wte = sd["language_model.transformer.wte.weight"]
    x = torch.cat((prefix, wte[tokens.clamp_min(0)].cuda()), dim=1)
    logits = ge.logits(x).cpu()
    with torch.no_grad():
        ref_rb = O.clipcap_logits(sd, tokens.clamp_min(0), embeds, cfg=cfg, rb=rb)
        pre_rb = O.mapper_forward(sd, embeds, projection_length=dims["P"], num_heads=dims["H"], num_layers=dims["N"],
                                  pre="transformer_mapper.", rb=rb)
    e_pre_rb = float((prefix.cpu() - pre_rb).abs().max())
    e_pre_32 = float((prefix.cpu() - torch.from_numpy(g["prefix"])).abs().max())
    e_rb = float(((logits - ref_rb) * valid[:, :, None]).abs().max())
    cols = sample_idx(V, 1024)
    e_32 = max(float(((logits[:, :, cols] - torch.from_numpy(g["logits.cols"])) * valid[:, :, None]).abs().max()),
               float(((logits[:, rows, :] - torch.from_numpy(g["logits.rows"])) * valid[:, rows, None]).abs().max()))
    drift = max(float(((ref_rb[:, :, cols] - torch.from_numpy(g["logits.cols"])) * valid[:, :, None]).abs().max()), 1e-9)
    print(f"{name}: prefix |max| {float(torch.from_numpy(g['prefix']).abs().max()):.2f}: vs oracle({pts}) {e_pre_rb:.3e}, vs reference fp32 "
          f"{e_pre_32:.3e};  logits |max| {float(g['logits.absmax']):.2f}: vs oracle({pts}) {e_rb:.3e}, vs reference fp32 {e_32:.3e} "
          f"({pts} oracle vs reference fp32: {drift:.3e})")
    pscale = float(torch.from_numpy(g["prefix"]).abs().max())
    assert e_pre_rb <= tol["prefix_rb"] * pscale and e_pre_32 <= tol["prefix_32"] * pscale
    assert e_rb <= tol["logits_rb"] and e_32 <= tol["logits_32"]
    assert e_32 <= 2.0 * drift + 1e-3          # no further from the reference than the rounding points themselves put the oracle
    out = dict(e_rb=e_rb, e_32=e_32, drift=drift, e_pre_rb=e_pre_rb, e_pre_32=e_pre_32, logits=logits, ref_rb=ref_rb, valid=valid, sd=sd,
               cfg=cfg, tokens=tokens, embeds=embeds, ge=ge, dims=dims, golden=g)

    # ---- training step: loss + gradients (model.py:94-113) ----
    eng.zero_grad()
    loss = float(eng.forward_backward(tokens.cuda(), embeds.cuda()))
    train = [k for k in sd if dims["full"] or k.startswith("transformer_mapper.")]
    sdr = {k: (v.clone().requires_grad_(True) if k in train else v) for k, v in sd.items()}
    ref_loss = O.clipcap_loss(sdr, tokens, embeds, cfg=cfg, rb=rb)
    ref_loss.backward()
    print(f"{name}: loss {loss:.6f}; oracle({pts}) {float(ref_loss):.6f}; reference fp32 {float(g['loss']):.6f}")
    assert abs(loss - float(ref_loss)) <= tol["loss_rb"] and abs(loss - float(g["loss"])) <= tol["loss_32"]
    unscale = 1.0 / float(eng.scaler.scale) if eng.scaler is not None else 1.0
    gm, gg = me.views(me.arena.g32), (ge.views(ge.arena.g32) if dims["full"] else {})
    worst_rb, worst_32, n = ("", 0.0), ("", 0.0), 0
    for k in train:
        mine = gm[k[len("transformer_mapper."):]] if k.startswith("transformer_mapper.") else gg[k[len("language_model."):]]
        mine = mine.cpu() * unscale
        if k.endswith("wte.weight"):
            mine = mine[:V]
        r = _rel(mine, sdr[k].grad)
        worst_rb = max(worst_rb, (k, r), key=lambda t: t[1])
        assert r <= tol["grad_rb"], (k, r)
        key = "grad0." + k
        nrm, smp = sampled(mine)
        ref_n, ref_s = float(g[key + ".norm"]), g[key + ".sample"]
        rs = float(np.linalg.norm(smp - ref_s) / max(np.linalg.norm(ref_s), 1e-20))
        worst_32 = max(worst_32, (k, rs), key=lambda t: t[1])
        assert abs(nrm - ref_n) <= tol["grad_32"] * ref_n, (k, nrm, ref_n)
        assert rs <= tol["grad_32"], (k, rs)
        n += 1
    out.update(grad_rb=worst_rb[1], grad_32=worst_32[1], loss=loss)
    print(f"{name}: {n} gradient tensors; worst rel. L2 vs oracle({pts}) {worst_rb[1]:.3e} ({worst_rb[0]}); worst vs reference fp32 "
          f"(strided sample) {worst_32[1]:.3e} ({worst_32[0]})")
    return out


# bf16-operand bars = 1.3 x the values measured on MI355X (printed by the tests with -s; round 4), so that a regression
# that doubled an error fails.  prefix_* are relative to max|prefix|.  loss_rb (the loss against the bf16-points oracle) is 1e-3 since round 5: the
# wave reductions changed their summation order (DPP instead of the xor butterfly), which moves the like-for-like figure inside the bf16 flip
# noise (config 4: 4.5e-4 -> 7.3e-4) while the figure against the reference's fp32 loss — the bar that matters, loss_32 — stayed at 1.3e-4.
TOL_C2 = dict(prefix_rb=2.3e-3, prefix_32=3.0e-3, logits_rb=2.1e-2, logits_32=1.9e-2, loss_rb=1e-3, loss_32=5e-4, grad_rb=5.7e-2, grad_32=7.7e-2)
#   measured: prefix 1.76e-3 / 2.29e-3 of max|prefix|, logits 1.60e-2 / 1.44e-2, loss 3e-5 / 2.8e-4, worst gradient 4.34e-2 / 5.86e-2
TOL_C3 = dict(prefix_rb=1.9e-3, prefix_32=2.9e-3, logits_rb=1.95e-2, logits_32=2.0e-2, loss_rb=1e-3, loss_32=5e-4, grad_rb=5.1e-2, grad_32=6.8e-2)
#   measured: prefix 1.41e-3 / 2.18e-3, logits 1.48e-2 / 1.54e-2, loss 4e-5 / 4e-6, worst gradient 3.88e-2 / 5.18e-2 (247 tensors)
TOL_C4 = dict(prefix_rb=2.5e-3, prefix_32=3.0e-3, logits_rb=2.6e-2, logits_32=2.3e-2, loss_rb=1e-3, loss_32=6e-4, grad_rb=5.4e-2, grad_32=7.0e-2)
#   measured: prefix 1.88e-3 / 2.29e-3, logits 1.96e-2 / 1.72e-2, loss 4.5e-4 / 1.5e-4, worst gradient 4.08e-2 / 5.36e-2 (391 tensors)


def test_config2_full_depth_logits_loss_grads():
    """BASELINE configs[1] architecture at full depth (8 + 12 layers), B=2, cap=40 with pads and an id-0 target."""
    _full_model_case("config2_full", TOL_C2)


def test_config3_full_depth_full_finetune_small():
    """BASELINE configs[2] architecture: the configs[1] mapper + 12-layer GPT-2-small, FULL finetune (the LM's weight gradients at
    GPT-2-small shapes, D=768 / 12 heads; round 4 fixture)."""
    _full_model_case("config3_full", TOL_C3)


def test_config4_full_depth_logits_loss_grads():
    """BASELINE configs[3] architecture: E=1024 -> D=1024 mapper (hd=128) + 24-layer GPT-2-medium (16 heads), full finetune."""
    _full_model_case("config4_full", TOL_C4)


def test_config4_full_size_step_properties():
    """configs[3] at its bench size (B=128, 24 layers, full finetune, dropout off): determinism, batch-permutation invariance and
    additivity of the loss and of the mapper / GPT-2 gradients — the properties tests/test_gpu_fullsize.py checks for configs[1]."""
    import bench
    c = dict(bench.CONFIGS["4"])
    me, ge, eng = bench.init_engines(c, torch.device("cuda", 0))
    gen = torch.Generator(device="cuda").manual_seed(17)
    embeds = torch.randn(c["B"], c["E"], generator=gen, device="cuda")
    tokens = torch.randint(1, c["V"], (c["B"], c["cap"]), generator=gen, device="cuda")
    tokens[::5, 33:] = -1
    tokens[2, 4] = 0

    def grads(t, e):
        eng.zero_grad()
        loss = eng.forward_backward(t, e)
        torch.cuda.synchronize()
        return float(loss), me.arena.g32.clone(), ge.arena.g32.clone(), float(eng.stats[1])

    l0, gm0, gg0, n0 = grads(tokens, embeds)
    assert torch.isfinite(gm0).all() and torch.isfinite(gg0).all() and abs(l0) < 20 and n0 == float((tokens > 0).sum())
    l0b, gm0b, gg0b, _ = grads(tokens, embeds)
    assert l0b == l0 and _rel(gm0b, gm0) <= 1e-5 and _rel(gg0b, gg0) <= 1e-5
    perm = torch.randperm(c["B"], device="cuda")
    l1, gm1, gg1, n1 = grads(tokens[perm], embeds[perm])
    assert n1 == n0 and abs(l1 - l0) <= 2e-5 * abs(l0) and _rel(gm1, gm0) <= 3e-3 and _rel(gg1, gg0) <= 3e-3
    h = c["B"] // 2
    la, gma, gga, na = grads(tokens[:h], embeds[:h])
    lb, gmb, ggb, nb = grads(tokens[h:], embeds[h:])
    assert na + nb == n0 and abs((la * na + lb * nb) / n0 - l0) <= 2e-5 * abs(l0)
    assert _rel((gma * na + gmb * nb) / n0, gm0) <= 2e-2 and _rel((gga * na + ggb * nb) / n0, gg0) <= 2e-2


# ---------------------------------------------------------------- configs[4]: beam decode at GPT-2-medium size ---------------

def _medium_lm(n_layer, wte_scale=2.0, seed=4401, npos=128, precision=None, wpe_scale=1.0):
    from tests import seeded
    from clipcap_amd.model.gpt2 import GPT2LM
    D, n_head, V = 1024, 16, 50257
    gsd = seeded.state_dict(seeded.gpt2_shapes(D, n_layer, V, npos), seed)
    gsd["transformer.wte.weight"] = gsd["transformer.wte.weight"] * wte_scale
    if wpe_scale != 1.0:
        gsd["transformer.wpe.weight"] = gsd["transformer.wpe.weight"] * wpe_scale
    lm = GPT2LM(n_embd=D, n_layer=n_layer, n_head=n_head, vocab_size=V, n_positions=npos, precision=precision)
    lm.load_state_dict({k: torch.from_numpy(v) for k, v in gsd.items()}, strict=False)
    return lm.to("cuda"), gsd


def test_config5_kv_cache_equals_reforward_24_layers():
    """GPT-2-medium depth and width, 320 rows (64 prefixes x beam 5): the KV-cached incremental logits equal the full re-forward."""
    from clipcap_amd.engine import DecodeSession
    lm, _ = _medium_lm(24)
    ge = lm.engine
    torch.manual_seed(2)
    x = torch.randn(320, 14, 1024, device="cuda") * 0.3
    full = ge.logits(x[:8])                       # re-forward reference on 8 of the rows (the rest exercise the batched tiles)
    sess = DecodeSession(ge, 320, 32)
    l = sess.forward(x[:, :10]).clone()
    scale = max(1.0, full.abs().max().item())

    def close(a, b):
        d = (a - b).float()
        return d.abs().max().item() <= 1e-2 * scale and d.pow(2).mean().sqrt().item() <= 2.5e-3 * scale

    assert close(l[:8], full[:, 9])
    for t in range(10, 14):
        l = sess.forward(x[:, t:t + 1])
        assert close(l[:8], full[:, t]), t


def test_config5_batched_beam_equals_per_sample_24_layers():
    """64 prefixes x beam 5, GPT-2-medium (24 layers): every sample's best caption from the batched decode equals the caption the
    same sample gets when decoded alone (the reference's batch-1 contract, base.py:17), and stopped beams freeze."""
    from types import SimpleNamespace
    from clipcap_amd.inference.base import generate_beam_tokens
    lm, _ = _medium_lm(24)
    model = SimpleNamespace(language_model=lm)
    gen = torch.Generator(device="cuda").manual_seed(9)
    pref = torch.randn(64, 10, 1024, generator=gen, device="cuda") * 0.5
    toks, scores, lens = generate_beam_tokens(model, pref, 5, 12, 1.0, 50256)
    assert toks.shape[:2] == (64, 5) and torch.isfinite(scores).all()
    # Beam search is chaotic under bf16 noise (a near-tie for the 5th beam at an early step changes which captions survive): over
    # all 64 samples 61 are identical (measured).  Identical captions must carry the same score.
    same, checked, misses = 0, (0, 5, 11, 23, 31, 40, 52, 63), []
    for i in checked:
        t1, s1, l1 = generate_beam_tokens(model, pref[i:i + 1], 5, 12, 1.0, 50256)
        b, b1 = int(scores[i].argmax()), int(s1[0].argmax())
        n = int(l1[0, b1])
        if torch.equal(toks[i, b, :n], t1[0, b1, :n]) and int(lens[i, b]) == n:
            same += 1
            assert abs(float(scores[i, b]) - float(s1[0, b1])) <= 2e-2
        else:
            misses.append((i, toks[i, b, : int(lens[i, b])].clone(), t1[0, b1, :n].clone()))
    assert same >= len(checked) - 1, same
    # A tolerated mismatch must be a near-tie, not a defect: both captions are re-scored with the SPLIT-bf16 model (the reference's
    # precision, logits 3e-5 from fp32; tests/test_gpu_x3.py shows 64 / 64 there) — their length-normalised log-probabilities differ
    # by less than the bf16 logit noise accumulated over the caption.
    if misses:
        lm.set_precision(32)
        wte = lm.get_input_embeddings().weight.detach()
        for i, ca, cb in misses:
            sc = []
            for cap in (ca, cb):
                x = torch.cat((pref[i:i + 1], wte[cap.long()][None]), dim=1)
                lp = torch.log_softmax(lm.engine.logits(x)[0, 9:9 + cap.numel()], -1)
                sc.append(float(lp.gather(1, cap.long()[:, None]).sum()) / cap.numel())
            print(f"sample {i}: batched vs alone captions differ; split-bf16 scores {sc[0]:.5f} vs {sc[1]:.5f} (margin {abs(sc[0] - sc[1]):.2e})")
            assert abs(sc[0] - sc[1]) <= 3e-2, (i, sc)
        lm.set_precision("bf16")


def test_beam_medium_width_tokens_vs_reference_and_oracle():
    """D=1024 / 16 heads / V=50257, 4 layers: the product's beam-5 captions are token-exact against the like-for-like oracle (bf16
    rounding points, full re-forward per step) and against the REFERENCE's own captions (tests/golden/beam_medium.npz), including
    the runs whose beams stop on EOS."""
    from types import SimpleNamespace
    from clipcap_amd.inference.base import generate_beam_tokens
    g = load_golden("beam_medium")
    D, NL, n_head, V, NPOS, seed = [int(v) for v in g["cfg"]]
    lm, gsd = _medium_lm(NL, float(g["wte_scale"]), seed, NPOS)
    model = SimpleNamespace(language_model=lm)
    sd = {"language_model." + k: torch.from_numpy(v) for k, v in gsd.items()}
    exact_ref = 0
    for case in ("beam0a", "beam0b", "beam1a", "beam1b"):
        eos, entry, beam = [int(v) for v in g[case + ".meta"]]
        pref = torch.from_numpy(g[case + ".prefix"])
        toks, scores, lens = generate_beam_tokens(model, pref.cuda(), beam, entry, 1.0, eos)
        b = int(scores[0].argmax())
        mine = toks[0, b, : int(lens[0, b])].cpu().numpy()
        ot, osc, ol, order = O.generate_beam_tokens(sd, pref, n_head=n_head, n_layer=NL, beam_size=beam, entry_length=entry, stop_token=eos,
                                                    rb=True)
        want = ot[order[0]][: int(ol[order[0]])].numpy()
        assert np.array_equal(mine, want), (case, mine, want)
        exact_ref += int(np.array_equal(mine, g[case + ".best"]))
    assert exact_ref >= 3, exact_ref      # fp32 reference: exact unless a top-2 margin is below the bf16 noise


def test_beam_deep_24_layers_tokens_vs_reference_and_oracle():
    """configs[4] at GPT-2-medium DEPTH (24 layers; tests/golden/beam_deep.npz = the reference's generate_beam, inference/base.py:55-132,
    entry_length 12, two prefixes, each also with a stop token that freezes beams mid-way).  bf16 operands: token-exact against the
    like-for-like oracle (bf16 rounding points, full re-forward per step); split-bf16 operands (the reference's precision): token-exact
    against the REFERENCE's captions."""
    from types import SimpleNamespace
    from clipcap_amd.inference.base import generate_beam_tokens
    g = load_golden("beam_deep")
    D, NL, n_head, V, NPOS, seed = [int(v) for v in g["cfg"]]
    assert NL == 24
    lm, gsd = _medium_lm(NL, float(g["wte_scale"]), seed, NPOS)
    model = SimpleNamespace(language_model=lm)
    sd = {"language_model." + k: torch.from_numpy(v) for k, v in gsd.items()}
    cases = ("beam0a", "beam0b", "beam1a", "beam1b")

    def mine(case):
        eos, entry, beam = [int(v) for v in g[case + ".meta"]]
        toks, scores, lens = generate_beam_tokens(model, torch.from_numpy(g[case + ".prefix"]).cuda(), beam, entry, 1.0, eos)
        b = int(scores[0].argmax())
        return toks[0, b, : int(lens[0, b])].cpu().numpy()

    exact_ref = 0
    for case in cases:
        eos, entry, beam = [int(v) for v in g[case + ".meta"]]
        got = mine(case)
        exact_ref += int(np.array_equal(got, g[case + ".best"]))
        if case in ("beam0a", "beam1b"):       # the like-for-like oracle re-forwards 24 layers per step on the host: two runs keep the test short
            ot, osc, ol, order = O.generate_beam_tokens(sd, torch.from_numpy(g[case + ".prefix"]), n_head=n_head, n_layer=NL, beam_size=beam,
                                                        entry_length=entry, stop_token=eos, rb=True)
            want = ot[order[0]][: int(ol[order[0]])].numpy()
            assert np.array_equal(got, want), (case, got, want)
    print(f"beam_deep, bf16 operands: {exact_ref} / 4 captions equal the reference's")
    assert exact_ref >= 3, exact_ref
    lm.set_precision(32)
    for case in cases:
        got = mine(case)
        assert np.array_equal(got, g[case + ".best"]), (case, got, g[case + ".best"])
    lm.set_precision("bf16")


def test_beam_varied_24_layers_ranking_under_competition():
    """tests/golden/beam_varied.npz: the reference's generate_beam (inference/base.py:55-132) on the seeded 24-layer GPT-2-medium with the
    position embeddings x8, wte x0.5 and temperature 1.5 — captions that do NOT repeat (>= 8 distinct tokens of 12, asserted by the
    generator), i.e. beam ranking with real competition at full depth (beam_deep's captions repeat one token).  split-bf16 operands (the
    reference's precision): token-exact against the reference; bf16 operands: token-exact against the like-for-like oracle."""
    from types import SimpleNamespace
    from clipcap_amd.inference.base import generate_beam_tokens
    g = load_golden("beam_varied")
    D, NL, n_head, V, NPOS, seed = [int(v) for v in g["cfg"]]
    temp = float(g["temperature"])
    assert NL == 24
    lm, gsd = _medium_lm(NL, float(g["wte_scale"]), seed, NPOS, wpe_scale=float(g["wpe_scale"]))
    assert np.array_equal(__import__("tests.seeded", fromlist=["checksum"]).checksum(gsd), g["param_checksum"])
    model = SimpleNamespace(language_model=lm)
    sd = {"language_model." + k: torch.from_numpy(v) for k, v in gsd.items()}
    cases = ("beam0a", "beam0b", "beam1a", "beam1b")
    for case in cases:
        assert len(set(int(t) for t in g[case[:-1] + "a.best"])) >= 8

    def mine(case):
        eos, entry, beam = [int(v) for v in g[case + ".meta"]]
        toks, scores, lens = generate_beam_tokens(model, torch.from_numpy(g[case + ".prefix"]).cuda(), beam, entry, temp, eos)
        b = int(scores[0].argmax())
        return toks[0, b, : int(lens[0, b])].cpu().numpy()

    exact_ref = 0
    for case in cases:
        eos, entry, beam = [int(v) for v in g[case + ".meta"]]
        got = mine(case)
        exact_ref += int(np.array_equal(got, g[case + ".best"]))
        if case in ("beam0a", "beam1b"):       # the like-for-like oracle re-forwards 24 layers per step on the host: two runs keep the test short
            ot, osc, ol, order = O.generate_beam_tokens(sd, torch.from_numpy(g[case + ".prefix"]), n_head=n_head, n_layer=NL, beam_size=beam,
                                                        entry_length=entry, stop_token=eos, temperature=temp, rb=True)
            want = ot[order[0]][: int(ol[order[0]])].numpy()
            assert np.array_equal(got, want), (case, got, want)
    print(f"beam_varied, bf16 operands: {exact_ref} / 4 captions equal the reference's")
    lm.set_precision(32)
    for case in cases:
        got = mine(case)
        assert np.array_equal(got, g[case + ".best"]), (case, got, g[case + ".best"])
    lm.set_precision("bf16")


def test_beam_number_to_generate_rounds_vs_reference():
    """number_to_generate > 1 (inference/base.py:79-130): the entry_length loop re-entered with live state — beams that never stopped keep
    growing (6 / 12 / 18 tokens), stopped beams freeze, scores are divided by seq_lengths once more per round.  tests/golden/beam_multi.npz =
    the reference's three generations for 2 prefixes x {stop never seen, stop mid-way, a stop every beam runs into}.  split-bf16 operands:
    every generation token-exact against the reference; bf16 operands: token-exact against the like-for-like oracle."""
    from types import SimpleNamespace
    from clipcap_amd.inference.base import generate_beam, generate_beam_rounds
    g = load_golden("beam_multi")
    D, NL, n_head, V, NPOS, seed = [int(v) for v in g["cfg"]]
    lm, gsd = _medium_lm(NL, float(g["wte_scale"]), seed, NPOS)
    model = SimpleNamespace(language_model=lm)
    sd = {"language_model." + k: torch.from_numpy(v) for k, v in gsd.items()}
    cases = [f"multi{i}{c}" for i in range(2) for c in "abc"]

    def mine(case):
        eos, entry, beam, ng = [int(v) for v in g[case + ".meta"]]
        out = []
        for toks, scores, lens in generate_beam_rounds(model, torch.from_numpy(g[case + ".prefix"]).cuda(), beam, entry, 1.0, eos, ng):
            b = int(scores[0].argmax())
            out.append(toks[0, b, : int(lens[0, b])].cpu().numpy())
        return out

    exact = total = 0
    for case in cases:
        eos, entry, beam, ng = [int(v) for v in g[case + ".meta"]]
        got = mine(case)
        assert len(got) == ng
        res = O.generate_beam_tokens(sd, torch.from_numpy(g[case + ".prefix"]), n_head=n_head, n_layer=NL, beam_size=beam, entry_length=entry,
                                     stop_token=eos, rb=True, rounds=ng)
        for r in range(ng):
            ot, osc, ol, order = res[r]
            want = ot[order[0]][: int(ol[order[0]])].numpy()
            assert np.array_equal(got[r], want), (case, r, got[r], want)
            exact += int(np.array_equal(got[r], g[f"{case}.gen{r}"]))
            total += 1
    print(f"beam_multi, bf16 operands: {exact} / {total} generations equal the reference's")
    assert exact >= total - 3, (exact, total)

    class Tok:                                   # the reference-signature entry point returns one text per generation
        eos_token = "<eos>"
        def __init__(self, e): self.e = e
        def encode(self, s): return [self.e]
        def decode(self, ids): return " ".join(str(int(i)) for i in ids)

    lm.set_precision(32)
    for case in cases:
        eos, entry, beam, ng = [int(v) for v in g[case + ".meta"]]
        got = mine(case)
        for r in range(ng):
            assert np.array_equal(got[r], g[f"{case}.gen{r}"]), (case, r, got[r], g[f"{case}.gen{r}"])
    case = cases[2]
    eos, entry, beam, ng = [int(v) for v in g[case + ".meta"]]
    texts = generate_beam(model, Tok(eos), torch.from_numpy(g[case + ".prefix"]).cuda(), number_to_generate=ng, beam_size=beam, entry_length=entry)
    assert texts == [" ".join(str(int(t)) for t in g[f"{case}.gen{r}"]) for r in range(ng)]
    lm.set_precision("bf16")


def test_windowed_mapper_at_real_sequence_length():
    """TransformerMapperWindowed at the reference's default window (window_size 16 -> 17 windows, model.py:28) and
    config-2 width: sequence 17*10 + 10 = 180 rows per sample, D=768, H=8 (hd 96) — the MFMA attention kernels over six 32-row blocks,
    forward and backward, against the like-for-like oracle (2 layers, B=2)."""
    from tests import seeded
    from clipcap_amd.engine import MapperEngine
    E, D, P, L, H, N, W, B = 512, 768, 10, 10, 8, 2, 17, 2
    msd = seeded.state_dict(seeded.mapper_shapes(E, D, P, L, N, W=W, use_pos=True), 4501)
    sd = {k: torch.from_numpy(v) for k, v in msd.items()}
    sd["pos_embeddings"] = sd["pos_embeddings"] * 0.1
    eng = MapperEngine(E, D, L, P, H, N, window=W, use_pos=True, device="cuda")
    for k, v in eng.views(eng.arena.w32).items():
        v.copy_(sd[k])
    x = torch.randn(B, W, E, generator=torch.Generator().manual_seed(3))
    out = eng.forward(x.cuda(), save=True)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.mapper_forward(sdr, x, projection_length=P, num_heads=H, num_layers=N, window=W, rb=True)
    with torch.no_grad():
        ref32 = O.mapper_forward(sd, x, projection_length=P, num_heads=H, num_layers=N, window=W)
    scale = float(ref32.abs().max())
    e_rb, e_32 = float((out.cpu() - ref.detach()).abs().max()), float((out.cpu() - ref32).abs().max())
    print(f"windowed mapper S=180: |out|max {scale:.2f}; vs oracle(bf16 points) {e_rb:.3e}; vs fp32 oracle {e_32:.3e}")
    assert e_rb <= 4e-3 * scale and e_32 <= 1e-2 * scale
    ref.square().mean().backward()
    eng.arena.grads().zero_()
    eng.backward(2.0 * out / out.numel())
    gv = eng.views(eng.arena.g32)
    worst = ("", 0.0)
    for k in sd:
        r = _rel(gv[k].cpu(), sdr[k].grad)
        worst = max(worst, (k, r), key=lambda t: t[1])
        assert r <= 5e-2, (k, r)
    print(f"windowed mapper S=180: worst relative gradient error {worst[1]:.3e} ({worst[0]})")
    atts = eng.attention_probs(B)
    assert atts[0].shape == (B, 180, 180, H) and float((atts[0].sum(dim=2) - 1).abs().max()) <= 1e-5


def test_reference_cli_default_shapes_gpt2_xl_width():
    """The reference's CLI defaults (clipcap/model/args.py, train/args.py): --language-model gpt2-xl (D=1600, 25 heads of 64),
    mapper heads 8 -> head dim 200 (no MFMA attention kernel: the LDS / VALU path), prefix = projection = 10; CLIP ViT-L/14
    embeddings (E=768).  2 of 48 GPT-2 layers, 2 of 8 mapper layers, B=2, cap=24: loss, logits and mapper gradients against the
    like-for-like oracle."""
    from tests import seeded
    E, D, P, L, H, N, n_head, NL, V, NPOS, cap, B = 768, 1600, 10, 10, 8, 2, 25, 2, 50257, 64, 24, 2
    gsd = seeded.state_dict(seeded.gpt2_shapes(D, NL, V, NPOS), 4601)
    msd = seeded.state_dict(seeded.mapper_shapes(E, D, P, L, N), 4602)
    sd = {"language_model." + k: torch.from_numpy(v) for k, v in gsd.items()}
    sd.update({"transformer_mapper." + k: torch.from_numpy(v) for k, v in msd.items()})
    cfg = dict(projection_length=P, prefix_length=L, heads=H, layers=N, n_head=n_head, n_layer=NL)
    dims = dict(E=E, D=D, P=P, L=L, H=H, N=N, n_head=n_head, NL=NL, V=V, NPOS=NPOS, full=False)
    me, ge, eng = _engines(sd, dims)
    gen = torch.Generator().manual_seed(5)
    tokens = torch.randint(1, V, (B, cap), generator=gen)
    tokens[1, cap - 5:] = -1
    embeds = torch.randn(B, E, generator=gen)
    eng.zero_grad()
    loss = float(eng.forward_backward(tokens.cuda(), embeds.cuda()))
    train = [k for k in sd if k.startswith("transformer_mapper.")]
    sdr = {k: (v.clone().requires_grad_(True) if k in train else v) for k, v in sd.items()}
    ref = O.clipcap_loss(sdr, tokens, embeds, cfg=cfg, rb=True)
    ref.backward()
    with torch.no_grad():
        ref32 = float(O.clipcap_loss(sd, tokens, embeds, cfg=cfg))
    print(f"gpt2-xl width: loss {loss:.6f}; oracle(bf16 points) {float(ref):.6f}; fp32 oracle {ref32:.6f}")
    assert abs(loss - float(ref)) <= 2e-3 and abs(loss - ref32) <= 3e-3
    gm = me.views(me.arena.g32)
    worst = ("", 0.0)
    for k in train:
        r = _rel(gm[k[len("transformer_mapper."):]].cpu(), sdr[k].grad)
        worst = max(worst, (k, r), key=lambda t: t[1])
        assert r <= 6e-2, (k, r)
    print(f"gpt2-xl width: worst relative mapper-gradient error {worst[1]:.3e} ({worst[0]})")
    # KV-cached decode at this width
    from clipcap_amd.engine import DecodeSession
    x = torch.randn(5, 13, D, generator=gen).cuda() * 0.3
    full = ge.logits(x)
    sess = DecodeSession(ge, 5, 16)
    l = sess.forward(x[:, :10]).clone()
    scale = max(1.0, float(full.abs().max()))
    assert float((l - full[:, 9]).abs().max()) <= 8e-3 * scale
    for t in range(10, 13):
        assert float((sess.forward(x[:, t:t + 1]) - full[:, t]).abs().max()) <= 8e-3 * scale, t


def test_full_finetune_gpt2_xl_width_grouped_weight_gradients_fit_their_scratch():
    """GPT-2-xl width, FULL finetune: a layer's four deferred weight gradients need 12 D^2 fp32 = 123 MB of slabs at D = 1600 — more
    than the 96 MiB weight-gradient scratch, so the group must flush early instead of writing past it.  K = B T = 1024
    rows (the grouped path's minimum), 1 GPT-2 layer, small vocabulary; every GPT-2 and mapper gradient against the oracle."""
    from tests import seeded
    E, D, P, L, H, N, n_head, NL, V, NPOS, cap, B = 64, 1600, 10, 10, 8, 1, 25, 1, 1000, 64, 22, 32
    gsd = seeded.state_dict(seeded.gpt2_shapes(D, NL, V, NPOS), 4701)
    msd = seeded.state_dict(seeded.mapper_shapes(E, D, P, L, N), 4702)
    sd = {"language_model." + k: torch.from_numpy(v) for k, v in gsd.items()}
    sd.update({"transformer_mapper." + k: torch.from_numpy(v) for k, v in msd.items()})
    cfg = dict(projection_length=P, prefix_length=L, heads=H, layers=N, n_head=n_head, n_layer=NL)
    dims = dict(E=E, D=D, P=P, L=L, H=H, N=N, n_head=n_head, NL=NL, V=V, NPOS=NPOS, full=True)
    me, ge, eng = _engines(sd, dims)
    gen = torch.Generator().manual_seed(6)
    tokens = torch.randint(1, V, (B, cap), generator=gen)
    embeds = torch.randn(B, E, generator=gen)
    eng.zero_grad()
    loss = float(eng.forward_backward(tokens.cuda(), embeds.cuda()))
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.clipcap_loss(sdr, tokens, embeds, cfg=cfg, rb=True)
    ref.backward()
    assert abs(loss - float(ref)) <= 2e-3, (loss, float(ref))
    gm, gg = me.views(me.arena.g32), ge.views(ge.arena.g32)
    worst = ("", 0.0)
    for k in sd:
        mine = gm[k[len("transformer_mapper."):]] if k.startswith("transformer_mapper.") else gg[k[len("language_model."):]]
        mine = mine.cpu()[:V] if k.endswith("wte.weight") else mine.cpu()
        r = _rel(mine, sdr[k].grad)
        worst = max(worst, (k, r), key=lambda t: t[1])
        assert r <= 8e-2, (k, r)
    print(f"gpt2-xl width full finetune: loss {loss:.5f} (oracle {float(ref):.5f}); worst relative gradient error {worst[1]:.3e} ({worst[0]})")
