"""Split-bf16 operand mode (CC_OP_BF16X3): the reference's DEFAULT precision, ``--fp-precision 32`` (clipcap/train/args.py:30-34 ->
``pl.Trainer(precision=...)``, train/train.py:82), on a chip without fp32 matrix cores: every GEMM product runs as three bf16 MFMA
terms hi*hi + hi*lo + lo*hi with fp32 accumulation, activations between kernels and attention are fp32.

north_star: "caption logits within 1e-3 of reference".  Asserted here, at FULL depth, against the reference's own fp32 outputs
(tests/golden/config2_full: 8-layer mapper + 12-layer GPT-2-small; config4_full: E=1024 mapper + 24-layer GPT-2-medium, full
finetune) — plus the gradients behind the mapper's ReLU (3.5-4.8 % off with bf16 operands), the beam search's batch-1 contract on
all 64 prefixes, and the training loop in this mode.
"""
import numpy as np
import pytest
import torch

from oracle import clipcap_oracle as O
from tests.test_gpu_configs import _full_model_case, _medium_lm
from tests.util import load_golden

pytestmark = pytest.mark.gpu

# measured (MI355X): logits 3e-5 .. 6e-5, prefix 2e-5 of its range, loss 1e-6, worst gradient tensor 1e-3 .. 3e-3
X3_TOL = dict(prefix_rb=2e-4, prefix_32=2e-4, logits_rb=1e-3, logits_32=1e-3, loss_rb=5e-5, loss_32=5e-5, grad_rb=1e-2, grad_32=1e-2)


def test_x3_config2_full_depth_logits_within_1e3_of_reference():
    """configs[1] architecture, 8 + 12 layers, B=2, cap=40: |logits - reference fp32 logits| <= 1e-3 on every loss-relevant row (the
    north-star bar, not a noise-floor argument), prefix, loss, and every mapper gradient tensor <= 1e-2 relative — including
    mlp.fc1 / norm2, which sit behind the ReLU mask."""
    r = _full_model_case("config2_full", X3_TOL, precision=32)
    print(f"split-bf16 operands, config2 full depth: max |logits - reference fp32| = {r['e_32']:.3e} (bar 1e-3; bf16 operands 1.5e-2, "
          f"fp16 operands 2.0e-3); worst gradient tensor vs reference {r['grad_32']:.3e}")
    assert r["e_32"] <= 1e-3 and r["grad_32"] <= 1e-2


def test_x3_config4_full_depth_medium_logits_within_1e3_of_reference():
    """configs[3] architecture: E=1024 -> D=1024 mapper (hd 128) + 24-layer GPT-2-medium, FULL finetune (391 gradient tensors)."""
    r = _full_model_case("config4_full", X3_TOL, precision=32)
    print(f"split-bf16 operands, config4 full depth: max |logits - reference fp32| = {r['e_32']:.3e}; worst gradient tensor vs reference "
          f"{r['grad_32']:.3e}")
    assert r["e_32"] <= 1e-3 and r["grad_32"] <= 1e-2


def test_x3_kv_cache_equals_reforward_24_layers():
    """GPT-2-medium, 320 rows: KV-cached incremental logits == full re-forward, to fp32-level agreement in this mode."""
    from clipcap_amd.engine import DecodeSession
    lm, _ = _medium_lm(24, precision=32)
    ge = lm.engine
    torch.manual_seed(2)
    x = torch.randn(320, 14, 1024, device="cuda") * 0.3
    full = ge.logits(x[:8])
    sess = DecodeSession(ge, 320, 32)
    l = sess.forward(x[:, :10]).clone()
    scale = max(1.0, full.abs().max().item())
    worst = float((l[:8] - full[:, 9]).abs().max())
    for t in range(10, 14):
        l = sess.forward(x[:, t:t + 1])
        worst = max(worst, float((l[:8] - full[:, t]).abs().max()))
    print(f"split-bf16 decode: KV cache vs re-forward max |diff| {worst:.3e} (scale {scale:.1f})")
    assert worst <= 2e-4 * scale


def _first_divergence(model, pref, i, beam, steps):
    """First step count after which sample i's beam SET differs between the batched and the single-sample decode, and the largest gap
    between a sequence only one of them kept and a sequence only the other kept (sum of log-probs) — (None, 0.0) if they never part."""
    from clipcap_amd.inference.base import generate_beam_tokens
    for n in range(1, steps + 1):
        tb, sb, lb = generate_beam_tokens(model, pref, beam, n, 1.0, 50256)
        ta, sa, la = generate_beam_tokens(model, pref[i:i + 1], beam, n, 1.0, 50256)
        kb = {tuple(tb[i, b].tolist()): float(sb[i, b] * lb[i, b]) for b in range(beam)}
        ka = {tuple(ta[0, b].tolist()): float(sa[0, b] * la[0, b]) for b in range(beam)}
        if set(kb) != set(ka):
            only_b = [v for k, v in kb.items() if k not in ka]
            only_a = [v for k, v in ka.items() if k not in kb]
            return n, max(abs(x - y) for x in only_b for y in only_a)
    return None, 0.0


def test_x3_batched_beam_equals_per_sample_all_64():
    """The reference's batch-1 contract (inference/base.py:17) on ALL 64 prefixes of configs[4] (GPT-2-medium, 24 layers, beam 5): the
    batched decode returns, for every sample, the caption that sample gets when decoded alone.  (bf16 operands: 61 / 64.)
    The batched (320-row) and the single-sample (5-row) steps run different GEMM tilings, i.e. different fp32 summation orders: their
    running sums of log-probs agree to 1e-5 .. 4e-5 over 12 steps (measured, tools/diag/x3_beam_divergence.py).  A sample may therefore part
    ways ONLY where two candidates for the last beam slot were closer than that noise — the test finds the step and checks the gap
    (round 5, after the wave reductions moved to DPP: sample 47 at step 6, candidates -33.305309 vs -33.305298) — and at most two may."""
    from types import SimpleNamespace
    from clipcap_amd.inference.base import generate_beam_tokens
    lm, _ = _medium_lm(24, precision=32)
    model = SimpleNamespace(language_model=lm)
    gen = torch.Generator(device="cuda").manual_seed(9)
    pref = torch.randn(64, 10, 1024, generator=gen, device="cuda") * 0.5
    toks, scores, lens = generate_beam_tokens(model, pref, 5, 12, 1.0, 50256)
    same, near_ties = 0, []
    for i in range(64):
        t1, s1, l1 = generate_beam_tokens(model, pref[i:i + 1], 5, 12, 1.0, 50256)
        b, b1 = int(scores[i].argmax()), int(s1[0].argmax())
        n = int(l1[0, b1])
        if torch.equal(toks[i, b, :n], t1[0, b1, :n]) and int(lens[i, b]) == n:
            same += 1
            assert abs(float(scores[i, b]) - float(s1[0, b1])) <= 1e-3
        else:
            step, gap = _first_divergence(model, pref, i, 5, 12)
            print(f"sample {i}: batched score {float(scores[i, b]):.6f} vs alone {float(s1[0, b1]):.6f}; beam sets part after step {step}, "
                  f"competing candidates {gap:.2e} apart")
            assert step is not None and gap <= 1e-4, (i, step, gap)          # a genuine near-tie, not a numerical defect
            near_ties.append(i)
    print(f"split-bf16 beam search: {same} / 64 batched captions identical to the per-sample decode; near-tie flips: {near_ties}")
    assert same >= 62


def test_x3_beam_medium_tokens_vs_reference():
    """beam-5 captions at GPT-2-medium width against the REFERENCE's own captions (tests/golden/beam_medium.npz): all four runs
    token-exact (bf16 operands: >= 3 of 4)."""
    from types import SimpleNamespace
    from clipcap_amd.inference.base import generate_beam_tokens
    g = load_golden("beam_medium")
    D, NL, n_head, V, NPOS, seed = [int(v) for v in g["cfg"]]
    lm, gsd = _medium_lm(NL, float(g["wte_scale"]), seed, NPOS, precision=32)
    model = SimpleNamespace(language_model=lm)
    for case in ("beam0a", "beam0b", "beam1a", "beam1b"):
        eos, entry, beam = [int(v) for v in g[case + ".meta"]]
        pref = torch.from_numpy(g[case + ".prefix"])
        toks, scores, lens = generate_beam_tokens(model, pref.cuda(), beam, entry, 1.0, eos)
        b = int(scores[0].argmax())
        mine = toks[0, b, : int(lens[0, b])].cpu().numpy()
        assert np.array_equal(mine, g[case + ".best"]), (case, mine, g[case + ".best"])


def test_x3_training_loop_through_the_model_api():
    """ClipCapModelPrefixOnly.set_precision(32) (what train() does for the reference's default --fp-precision): three optimizer steps
    through the fused trainer path lower the loss, the operand images follow the master weights (the step after an update sees the new
    weights), and the loss of step 0 equals the oracle's fp32 loss."""
    from tests import seeded
    from clipcap_amd.engine import ClipCapEngine, Gpt2Engine, MapperEngine
    E, D, P, L, H, N, n_head, NL, V, NPOS, cap, B = 64, 128, 4, 4, 4, 2, 4, 2, 300, 32, 8, 4
    gsd = seeded.state_dict(seeded.gpt2_shapes(D, NL, V, NPOS), 77)
    msd = seeded.state_dict(seeded.mapper_shapes(E, D, P, L, N), 78)
    sd = {"language_model." + k: torch.from_numpy(v) for k, v in gsd.items()}
    sd.update({"transformer_mapper." + k: torch.from_numpy(v) for k, v in msd.items()})
    me = MapperEngine(E, D, L, P, H, N, device="cuda", precision=32)
    ge = Gpt2Engine(D, n_head, NL, V, NPOS, device="cuda", precision=32)
    for pre, eng_ in (("transformer_mapper.", me), ("language_model.", ge)):
        for k, v in eng_.views(eng_.arena.w32).items():
            v.copy_(sd[pre + k])
    eng = ClipCapEngine(me, ge, train_lm=True)
    assert eng._scaler(torch.device("cuda", 0)) is None          # no loss scale in this mode
    gen = torch.Generator().manual_seed(5)
    tokens = torch.randint(1, V, (B, cap), generator=gen)
    tokens[1, cap - 3:] = -1
    embeds = torch.randn(B, E, generator=gen)
    cfg = dict(projection_length=P, prefix_length=L, heads=H, layers=N, n_head=n_head, n_layer=NL)
    with torch.no_grad():
        ref = float(O.clipcap_loss(sd, tokens, embeds, cfg=cfg))
    losses = []
    for step in range(1, 4):
        eng.zero_grad()
        losses.append(float(eng.forward_backward(tokens.cuda(), embeds.cuda())))
        eng.optimizer_step(1e-3, step)
    print(f"split-bf16 training: losses {losses}; fp32 oracle loss at step 0 {ref:.6f}")
    assert abs(losses[0] - ref) <= 2e-5
    assert losses[2] < losses[1] < losses[0]


def test_x3_sizes_and_refusals():
    """Buffer sizes of the mode (include/clipcap_hip.h OPERAND TYPE) and the entry points that do not exist in it."""
    import ctypes as C
    from clipcap_amd import _lib
    from clipcap_amd.engine import DecodeSession, Gpt2Engine
    ge = Gpt2Engine(128, 4, 2, 300, 32, device="cuda", precision=32)
    assert ge.arena.w16.numel() == 6 * ge.arena.n
    sess = DecodeSession(ge, 3, 16)
    assert sess.kv.dtype == torch.float32
    l = _lib.lib()
    assert l.cc_gpt2_transpose_weights(C.byref(ge.cfg), C.c_void_p(ge.arena.w16.data_ptr()), None) == -1
    assert l.cc_cast_op16(2, C.c_void_p(ge.arena.w32.data_ptr()), C.c_void_p(ge.arena.w16.data_ptr()), 8, None) == -1


def test_x3_full_finetune_with_dropout_matches_oracle_with_the_same_masks():
    """GPT-2 train-mode dropout in this mode (the fp32 attention kernels carry the attention-probability mask): loss and every
    gradient against the oracle run with the kernels' own masks, to this mode's tolerances."""
    from tests.test_gpu_dropout import _build, _mask
    E, D, P, L, H, N, n_head, n_layer, V, npos = 16, 128, 2, 3, 2, 1, 2, 2, 157, 32
    eng, sd, cfg = _build(E, D, P, L, H, N, n_head, n_layer, V, npos)
    eng.mapper.set_precision(32)
    eng.gpt2.set_precision(32)
    torch.manual_seed(1)
    B, cap = 3, 7
    tokens, embeds = torch.randint(1, V, (B, cap)), torch.randn(B, E)
    tokens[1, 5:] = -1
    T = L + cap
    p_e, p_a, p_r, seed = 0.1, 0.15, 0.2, 0x1234_5678_9abc
    loss = eng.forward_backward(tokens.cuda(), embeds.cuda(), dropout=(p_e, p_a, p_r, seed))
    drop = {"p_embd": p_e, "p_attn": p_a, "p_resid": p_r, "embd": _mask(seed, 0, 0, p_e, (B, T, D)),
            "attn": [_mask(seed, 1, l, p_a, (B, n_head, T, T)) for l in range(n_layer)],
            "resid_attn": [_mask(seed, 2, l, p_r, (B, T, D)) for l in range(n_layer)],
            "resid_mlp": [_mask(seed, 3, l, p_r, (B, T, D)) for l in range(n_layer)]}
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.clipcap_loss(sdr, tokens, embeds, cfg=cfg, drop=drop)
    ref.backward()
    assert abs(float(loss) - float(ref.detach())) <= 2e-5, (float(loss), float(ref.detach()))
    worst = ("", 0.0)
    for pre, e in (("transformer_mapper.", eng.mapper), ("language_model.", eng.gpt2)):
        for k, v in e.views(e.arena.g32).items():
            r = sdr[pre + k].grad
            if "lm_head" in k or r is None:
                continue
            if k.endswith("wte.weight"):
                v = v[:V]
            rel = ((v.cpu() - r).norm() / r.norm().clamp_min(1e-12)).item()
            worst = max(worst, (pre + k, rel), key=lambda t: t[1])
    print(f"split-bf16 operands, dropout: loss {float(loss):.6f} vs oracle {float(ref.detach()):.6f}; worst gradient {worst[1]:.3e} ({worst[0]})")
    assert worst[1] <= 2e-3, worst


def test_x3_windowed_mapper_at_real_sequence_length():
    """TransformerMapperWindowed at the reference's default window (17 x 10 + 10 = 180 rows per sample) in this mode: the one-wave-per-row
    fp32 attention kernels (the S x S tile does not fit the LDS kernels), forward and backward, against the fp32 oracle."""
    from tests import seeded
    from clipcap_amd.engine import MapperEngine
    E, D, P, L, H, N, W, B = 512, 768, 10, 10, 8, 2, 17, 2
    msd = seeded.state_dict(seeded.mapper_shapes(E, D, P, L, N, W=W, use_pos=True), 4501)
    sd = {k: torch.from_numpy(v) for k, v in msd.items()}
    sd["pos_embeddings"] = sd["pos_embeddings"] * 0.1
    eng = MapperEngine(E, D, L, P, H, N, window=W, use_pos=True, device="cuda", precision=32)
    for k, v in eng.views(eng.arena.w32).items():
        v.copy_(sd[k])
    x = torch.randn(B, W, E, generator=torch.Generator().manual_seed(3))
    out = eng.forward(x.cuda(), save=True)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.mapper_forward(sdr, x, projection_length=P, num_heads=H, num_layers=N, window=W)
    scale = float(ref.detach().abs().max())
    e_32 = float((out.cpu() - ref.detach()).abs().max())
    ref.square().mean().backward()
    eng.arena.grads().zero_()
    eng.backward(2.0 * out / out.numel())
    gv = eng.views(eng.arena.g32)
    worst = ("", 0.0)
    for k in sd:
        r = float((gv[k].cpu() - sdr[k].grad).norm() / sdr[k].grad.norm().clamp_min(1e-20))
        worst = max(worst, (k, r), key=lambda t: t[1])
    print(f"split-bf16 windowed mapper S=180: |out|max {scale:.2f}; vs fp32 oracle {e_32:.3e}; worst gradient {worst[1]:.3e} ({worst[0]})")
    assert e_32 <= 1e-4 * scale and worst[1] <= 5e-3
    atts = eng.attention_probs(B)
    assert atts[0].shape == (B, 180, 180, H) and float((atts[0].sum(dim=2) - 1).abs().max()) <= 1e-5


def test_x3_module_autograd_paths_vs_reference_gradients():
    """The reference-style autograd surface in this mode: ``lm(inputs_embeds=x).logits`` -> backward (cc_gpt2_logits_bwd) and
    ``TransformerMapper(x)`` -> backward (cc_mapper_bwd) against the reference's own gradients (tests/golden/gpt2_tiny, mapper_tiny),
    at fp32-level tolerances."""
    from clipcap_amd.model.gpt2 import GPT2LM
    from clipcap_amd.model.mapper import TransformerMapper
    from tests.util import sd_of
    g = load_golden("gpt2_tiny")
    D, n_layer, n_head, V, npos = [int(v) for v in g["cfg"]]
    lm = GPT2LM(n_embd=D, n_layer=n_layer, n_head=n_head, vocab_size=V, n_positions=npos, precision=32)
    lm.load_state_dict(sd_of(g), strict=False)
    lm = lm.to("cuda")
    x = torch.from_numpy(g["in.x"]).cuda().requires_grad_(True)
    logits = lm(inputs_embeds=x).logits
    assert (logits.detach().cpu() - torch.from_numpy(g["logits"])).abs().max().item() <= 1e-4
    logits.square().mean().backward()
    ref = torch.from_numpy(g["grad.in.x"])
    worst = float((x.grad.cpu() - ref).norm() / ref.norm())
    for k, p in lm.named_parameters():
        if "grad." + k in g and "lm_head" not in k:
            r = torch.from_numpy(g["grad." + k])
            worst = max(worst, float((p.grad.cpu() - r).norm() / r.norm().clamp_min(1e-12)))
    gm = load_golden("mapper_tiny")
    E, Dm, P, L, H, N, B = [int(v) for v in gm["dims"]]
    m = TransformerMapper(E, Dm, L, P, H, N, precision=32)
    m.load_state_dict(sd_of(gm))
    m = m.to("cuda")
    xin = torch.from_numpy(gm["in.x"]).cuda()
    out = m(xin)
    assert (out.detach().cpu() - torch.from_numpy(gm["out"])).abs().max().item() <= 1e-4
    out.square().mean().backward()
    wm = 0.0
    for k, p in m.named_parameters():
        if "grad." + k in gm:
            r = torch.from_numpy(gm["grad." + k])
            wm = max(wm, float((p.grad.cpu() - r).norm() / r.norm().clamp_min(1e-12)))
    print(f"split-bf16 autograd paths: GPT-2 worst relative gradient error {worst:.2e}, mapper {wm:.2e}")
    assert worst <= 2e-3 and wm <= 2e-3


@pytest.mark.parametrize("B,S,H,hd,causal", [(3, 50, 2, 64, 1), (2, 20, 3, 96, 0), (2, 64, 2, 64, 1), (2, 33, 1, 96, 1), (1, 32, 2, 64, 0),
                                             (2, 1, 2, 64, 1), (2, 7, 1, 128, 0), (1, 61, 2, 96, 0), (2, 75, 1, 64, 1), (1, 180, 2, 96, 0),
                                             (2, 17, 2, 32, 1), (3, 40, 2, 128, 1)])
def test_x3_attention_kernels_vs_fp32(B, S, H, hd, causal):
    """cc_attention_fwd / _bwd in the split-bf16 build (fp32 tensors): the three-term MFMA kernels (k_attn_fwd_mfma3 for head dim 64 / 96 /
    128; k_attn_bwd_m3 for head dim 64 / 96, S <= 64 — one and two 32-row blocks, ragged last block) and the fp32 VALU kernels they fall
    back to (head dim 32, head dim 128 backward, S = 75 / 180) against fp32 torch: output and lse to 2e-4, gradients to 5e-4 of their scale
    (every product carries ~16 mantissa bits per operand)."""
    from tests.test_gpu_kernels import _attn_ref, _lib, _p, _st
    torch.manual_seed(S * 7 + hd + causal)
    D = H * hd
    qkv = torch.randn(B * S, 3 * D, device="cuda")
    out = torch.full((B * S, D), float("nan"), device="cuda")
    lse = torch.empty(B, H, S, device="cuda")
    assert _lib().cc_attention_fwd(2, _p(qkv), B, S, H, hd, causal, _p(out), _p(lse), _st()) == 0
    qkv_r = qkv.clone().requires_grad_(True)
    ref, lse_ref = _attn_ref(qkv_r, B, S, H, hd, causal)
    torch.cuda.synchronize()
    assert (lse - lse_ref).abs().max().item() <= 2e-4
    assert (out.view(B, S, D) - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    dout = torch.randn(B * S, D, device="cuda")
    dqkv = torch.full_like(qkv, float("nan"))
    delta = torch.empty(B * H * S, device="cuda")
    assert _lib().cc_attention_bwd(2, _p(qkv), _p(dout), _p(out), _p(lse), _p(delta), B, S, H, hd, causal, _p(dqkv), _st()) == 0
    ref.backward(dout.view(B, S, D))
    torch.cuda.synchronize()
    g = qkv_r.grad
    assert torch.isfinite(dqkv).all()
    assert (dqkv - g).abs().max().item() <= 5e-4 * max(1.0, g.abs().max().item()), (dqkv - g).abs().max().item()
