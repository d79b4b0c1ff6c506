"""GPU tests of the module-level API (clipcap_amd.model / .inference / .train) — the calls a user of the reference makes."""
import os

import numpy as np
import pytest
import torch

from oracle import clipcap_oracle as O
from tests.test_api_surface import FakeTokenizer, _write_dataset
from tests.util import load_golden, sd_of

pytestmark = pytest.mark.gpu


def _model_from_train_fixture(mode="prefix_only"):
    from clipcap_amd.encoders import EncoderConfig
    from clipcap_amd.model import ClipCapModel, ClipCapModelPrefixOnly, Config, TrainingConfig
    from clipcap_amd.model.gpt2 import GPT2LM
    g = load_golden(f"train_{mode}")
    E, D, P, L, H, N, n_head, n_layer, V, npos = [int(v) for v in g["cfg"]]
    # the fixtures were captured with resid_pdrop = embd_pdrop = attn_pdrop = 0 (oracle/gen_golden.py)
    lm = GPT2LM(n_embd=D, n_layer=n_layer, n_head=n_head, vocab_size=V, n_positions=npos, embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    cfg = Config(language_model="unused", train_language_model=(mode == "full"), prefix_length=L, projection_length=P, transformer_layers=N,
                 transformer_attention_heads=H, encoder_config=EncoderConfig(encoder_embedding_size=E),
                 training_config=TrainingConfig(optimizer_lr=1e-3, use_deepspeed_optimisers=False, scheduler_warmup_steps=2, total_steps=6))
    m = (ClipCapModel if mode == "full" else ClipCapModelPrefixOnly)(cfg, language_model=lm)
    m.load_state_dict(sd_of(g), strict=True)
    return m.to("cuda"), g


def test_module_forward_logits_and_to_device():
    m, g = _model_from_train_fixture()
    assert m.transformer_mapper.engine.arena.w32.is_cuda and next(m.parameters()).is_cuda
    tokens = torch.from_numpy(g["in.tokens"])
    out = m(tokens.clamp_min(0).cuda(), torch.from_numpy(g["in.embeds"]).cuda(), tokens.ge(0).cuda())
    keep = np.concatenate([np.ones((tokens.shape[0], 3), bool), g["in.tokens"] >= 0], axis=1)
    err = np.abs(out.logits.detach().cpu().numpy()[keep] - g["logits0"][keep]).max()   # differentiable, like the reference's
    print("module forward: max |logits - reference fp32| =", err)
    assert out.logits.shape == g["logits0"].shape and err <= 3e-2
    pre = m.transformer_mapper(torch.from_numpy(g["in.embeds"]).cuda())
    assert pre.shape == (4, 3, 64)
    emb = m.language_model.get_input_embeddings()(torch.tensor([[1, 2]], device="cuda"))
    assert emb.shape == (1, 2, 64)


def test_input_embedding_autograd_runs_on_the_library_and_equals_torch():
    """`language_model.get_input_embeddings()(tokens)` with gradients enabled (a full finetune through Module.forward, reference
    model.py:44): gather and scatter-add gradient on cc_embed_tokens / cc_embed_tokens_bwd — values exact, the gradient equal to
    torch.nn.functional.embedding's (repeated ids accumulate; fp32 atomics in another order: 1e-6 of the row scale)."""
    m, g = _model_from_train_fixture("full")
    emb = m.language_model.get_input_embeddings()
    w = emb.weight
    assert w.requires_grad and w.is_cuda
    V, D = w.shape
    ids = torch.randint(0, V, (5, 7), generator=torch.Generator().manual_seed(5)).cuda()
    ids[0, :4] = 3                      # one id four times, and the last vocabulary row
    ids[4, 6] = V - 1
    up = torch.randn(5, 7, D, generator=torch.Generator().manual_seed(6)).cuda()
    out = emb(ids)
    assert out.requires_grad and torch.equal(out.detach(), w.detach()[ids])
    (gw,) = torch.autograd.grad((out * up).sum(), w)
    w2 = w.detach().clone().requires_grad_(True)
    (gr,) = torch.autograd.grad((torch.nn.functional.embedding(ids, w2) * up).sum(), w2)
    assert gw.shape == gr.shape == (V, D)
    assert torch.allclose(gw, gr, rtol=1e-6, atol=1e-6) and float(gw[3].abs().max()) > 0 and float(gw[V - 1].abs().max()) > 0
    with pytest.raises(IndexError):
        emb(torch.tensor([[0, V]]))     # host ids are validated like F.embedding


@pytest.mark.parametrize("mode", ["prefix_only", "full"])
def test_three_optimizer_steps_track_the_reference(mode):
    """fused_step x3 with the reference's lr schedule: losses follow the golden trajectory; parameters move along the
    reference's AdamW update (bf16 GEMM noise is amplified by Adam's normalisation, so the check is on direction)."""
    from clipcap_amd.model.optim import linear_warmup_decay
    m, g = _model_from_train_fixture(mode)
    m.train()
    tokens, embeds = torch.from_numpy(g["in.tokens"]).cuda(), torch.from_numpy(g["in.embeds"]).cuda()
    sched = linear_warmup_decay(2, 6)
    before = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    # host replay of torch.optim.AdamW (fp64) driven by the gradients the kernels produced: pins weight decay, bias correction, the
    # step counter and the schedule on the FULL fused path to 1e-6, independently of bf16 noise in the gradients themselves
    arenas = [m.transformer_mapper.engine.arena] + ([m.language_model.engine.arena] if mode == "full" else [])
    rep = [dict(p=a.w32.double().cpu(), m=torch.zeros(a.n, dtype=torch.float64), v=torch.zeros(a.n, dtype=torch.float64)) for a in arenas]
    losses = []
    for s_ in range(3):
        lr = 1e-3 * sched(s_)
        losses.append(float(m.fused_step((tokens.clone(), embeds), lr=lr)))
        for a, r in zip(arenas, rep):
            gq = a.g32.double().cpu()
            r["p"] = r["p"] * (1.0 - lr * 0.01)
            r["m"] = 0.9 * r["m"] + 0.1 * gq
            r["v"] = 0.999 * r["v"] + 0.001 * gq * gq
            r["p"] = r["p"] - (lr / (1 - 0.9 ** (s_ + 1))) * r["m"] / (r["v"].sqrt() / (1 - 0.999 ** (s_ + 1)) ** 0.5 + 1e-8)
            assert (a.w32.double().cpu() - r["p"]).abs().max().item() <= 2e-6, (mode, s_)
    print(mode, "losses", losses, "golden", g["losses"])
    assert np.abs(np.array(losses) - g["losses"]).max() <= 3e-2
    after = m.state_dict()
    cos, cos_strong = [], []
    for k in before:
        key = "sd_after3." + k
        if key in g and "lm_head" not in k:
            ours = (after[k].cpu() - before[k]).flatten()
            ref = (torch.from_numpy(g[key]) - before[k]).flatten()
            if ref.norm() > 0:
                cos.append(float(torch.dot(ours, ref) / (ours.norm() * ref.norm() + 1e-30)))
            # Adam moves every element by ~lr * sign(gradient): elements whose reference gradient is far above the bf16 noise of
            # ours must move exactly as the reference's do
            gk = "grad0." + k
            if gk in g and ref.norm() > 0:
                g0 = torch.from_numpy(g[gk]).flatten().abs()
                strong = g0 >= 0.5 * g0.pow(2).mean().sqrt()
                if int(strong.sum()) >= 8:
                    o, r = ours[strong], ref[strong]
                    cos_strong.append((float(torch.dot(o, r) / (o.norm() * r.norm() + 1e-30)), k))
    print(mode, "min/mean cosine of 3-step parameter deltas:", min(cos), sum(cos) / len(cos), "| elements with a clear gradient: min",
          min(cos_strong), "over", len(cos_strong), "tensors")
    assert sum(cos) / len(cos) >= 0.9
    assert min(cos_strong)[0] >= 0.98, min(cos_strong)
    if mode == "prefix_only":   # the LM must be untouched
        for k in before:
            if k.startswith("language_model."):
                assert torch.equal(after[k].cpu(), before[k]), k


def test_training_step_autograd_path_equals_fused_path():
    m, g = _model_from_train_fixture()
    m.train()
    tokens, embeds = torch.from_numpy(g["in.tokens"]).cuda(), torch.from_numpy(g["in.embeds"]).cuda()
    oc = m.configure_optimizers()
    opt, sch = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
    assert oc["lr_scheduler"]["interval"] == "step" and oc["lr_scheduler"]["frequency"] == 1
    opt.zero_grad()
    loss = m.training_step((tokens.clone(), embeds), 0)
    loss.backward()
    grads = {n: p.grad.clone() for n, p in m.transformer_mapper.named_parameters()}
    assert all(gr is not None and torch.isfinite(gr).all() for gr in grads.values())
    opt.step()
    sch.step()
    a = {k: v.clone() for k, v in m.transformer_mapper.state_dict().items()}
    m2, _ = _model_from_train_fixture()
    m2.train()
    m2.fused_step((tokens.clone(), embeds), lr=0.0)   # warm-up factor at step 0 is 0 (model.py:79-83)
    for k, v in m2.transformer_mapper.state_dict().items():
        assert torch.allclose(v, a[k], atol=1e-7), k


def test_kv_cached_decode_equals_full_forward():
    from clipcap_amd.engine import DecodeSession
    m, g = _model_from_train_fixture()
    eng = m.language_model.engine
    torch.manual_seed(0)
    x = torch.randn(3, 9, 64, device="cuda") * 0.5
    full = eng.logits(x)                                  # (3, 9, V)
    sess = DecodeSession(eng, 3, 16)
    l0 = sess.forward(x[:, :4])
    assert (l0 - full[:, 3]).abs().max().item() <= 2e-3
    for t in range(4, 9):
        lt = sess.forward(x[:, t:t + 1])
        assert (lt - full[:, t]).abs().max().item() <= 2e-3, t
    # fan-out (copy) then in-place ancestry reorder keep every row's history
    src = torch.tensor([2, 2, 0, 1], dtype=torch.int32, device="cuda")
    s2 = sess.expand(src, 4)
    assert s2.R == 4 and s2.pos == 9
    x2 = torch.randn(4, 2, 64, device="cuda") * 0.5
    l9 = s2.forward(x2[:, :1]).clone()   # the logits buffer is reused by the next forward()
    perm = torch.tensor([1, 0, 3, 3], dtype=torch.int32, device="cuda")
    s2.reorder(perm)
    l10 = s2.forward(x2[:, 1:2])
    for r in range(4):
        hist = torch.cat((x[int(src[r])], x2[r, :1]))[None]
        assert (l9[r] - eng.logits(hist)[0, -1]).abs().max().item() <= 2e-3
        pr = int(perm[r])
        hist2 = torch.cat((x[int(src[pr])], x2[pr, :1], x2[r, 1:2]))[None]
        assert (l10[r] - eng.logits(hist2)[0, -1]).abs().max().item() <= 2e-3, r


def test_generate_beam_matches_reference_tokens():
    """Token-exact against the reference's generate_beam on the golden beam fixtures (incl. early-EOS and frozen beams)."""
    from clipcap_amd.inference import generate_beam, generate_beam_tokens
    from clipcap_amd.model.gpt2 import GPT2LM
    from types import SimpleNamespace
    g = load_golden("beam_tiny")
    D, n_layer, n_head, V, npos = [int(v) for v in g["cfg"]]
    lm = GPT2LM(n_embd=D, n_layer=n_layer, n_head=n_head, vocab_size=V, n_positions=npos)
    lm.load_state_dict(sd_of(g), strict=False)
    model = SimpleNamespace(language_model=lm.to("cuda"))
    cases = [str(int(c)) for c in g["cases"]] + ["T"]
    osd = {"language_model." + k: v for k, v in sd_of(g).items()}
    exact_vs_reference = 0
    for c in cases:
        eos, entry, beam = [int(v) for v in g[f"beam{c}.meta"]]
        temp = 0.7 if c == "T" else 1.0
        pref = torch.from_numpy(g[f"beam{c}.prefix"])
        toks, scores, lens = generate_beam_tokens(model, pref.cuda(), beam, entry, temp, eos)
        b = int(scores[0].argmax())
        best = toks[0, b, : int(lens[0, b])].cpu().numpy()
        # like-for-like: the oracle with the kernels' bf16 rounding points must give the same tokens, always
        ot, osc, ol, oo = O.generate_beam_tokens(osd, pref, n_head=n_head, n_layer=n_layer, beam_size=beam, entry_length=entry,
                                                 temperature=temp, stop_token=eos, rb=True)
        obest = ot[oo[0]][: int(ol[oo[0]])].numpy()
        assert np.array_equal(best, obest), (c, best, obest)
        # vs the reference's own fp32 run: exact whenever its two best beams are separated by more than bf16 noise
        ft, fsc, fl, fo = O.generate_beam_tokens(osd, pref, n_head=n_head, n_layer=n_layer, beam_size=beam, entry_length=entry,
                                                 temperature=temp, stop_token=eos)
        margin = float(fsc[fo[0]] - fsc[fo[1]])
        ok = np.array_equal(best, g[f"beam{c}.best"])
        print(f"beam case {c}: ours {best.tolist()} ref {g[f'beam{c}.best'].tolist()} fp32 top-2 margin {margin:.2e} {'OK' if ok else 'differs'}")
        if margin > 5e-3:
            assert ok, c
        exact_vs_reference += ok
        tk = FakeTokenizer(V, eos)
        txt = generate_beam(model, tk, pref.cuda(), beam_size=beam, entry_length=entry, temperature=temp)
        assert txt == [tk.decode(best)]
    assert exact_vs_reference >= len(cases) - 1
    # batched decode == per-sample decode
    prefs = torch.cat([torch.from_numpy(g[f"beam{c}.prefix"]) for c in ("1", "2", "3")]).cuda()
    toks, scores, lens = generate_beam_tokens(model, prefs, 5, 12, 1.0, 50)
    for i, c in enumerate(("1", "2", "3")):
        t1, s1, l1 = generate_beam_tokens(model, prefs[i:i + 1], 5, 12, 1.0, 50)
        n = min(t1.shape[2], toks.shape[2])
        b = int(s1[0].argmax())
        assert int(scores[i].argmax()) == b
        assert torch.equal(toks[i, b, : int(l1[0, b])], t1[0, b, : int(l1[0, b])])


def test_train_driver_end_to_end(tmp_path):
    """python -m clipcap_amd.train equivalent on a tiny on-disk dataset: config + checkpoints written, loss decreases,
    checkpoint loads through clipcap_amd.load and decodes."""
    import argparse
    import clipcap_amd
    from clipcap_amd.inference import generate_beam
    from clipcap_amd.model import add_model_args
    from clipcap_amd.model.gpt2 import GPT2LM
    from clipcap_amd.train import add_training_args, train
    _write_dataset(tmp_path / "ds", n=48, E=24, shards=(20, 28))
    lm = GPT2LM(n_embd=64, n_layer=2, n_head=4, vocab_size=157, n_positions=96)
    lm.save_pretrained(str(tmp_path / "lm"))
    args = add_model_args(add_training_args(argparse.ArgumentParser())).parse_args([
        "--input-dataset", str(tmp_path / "ds"), "--output-folder", str(tmp_path / "out"), "--language-model", str(tmp_path / "lm"),
        "--batch-size", "16", "--epochs", "3", "--optimizer-lr", "2e-3", "--scheduler-warmup-steps", "2", "--checkpoint-filename-prefix",
        "t", "--prefix-length", "4", "--projection-length", "4", "--transformer-layers", "2", "--transformer-attention-heads", "4",
        "--logging-frequency", "1000"])
    tok = FakeTokenizer()
    assert train(args, tokenizer=tok) == 0
    files = sorted(os.listdir(tmp_path / "out"))
    assert files == ["t_config.yaml", "t_epoch_0.ckpt", "t_epoch_1.ckpt", "t_epoch_2.ckpt", "t_final.ckpt"]
    model, _ = clipcap_amd.load(str(tmp_path / "out" / "t_final.ckpt"), str(tmp_path / "out" / "t_config.yaml"), device="cuda",
                                from_checkpoint=True, tokenizer=tok)
    prefix = model.transformer_mapper(torch.randn(1, 24, device="cuda"))
    text = generate_beam(model, tok, prefix, beam_size=3, entry_length=6)
    assert isinstance(text, list) and len(text) == 1 and isinstance(text[0], str)


def test_kv_cached_decode_split_k_path_equals_full_forward():
    """Wider model (D=256: K = 256 / 1024) so that the skinny split-K GEMMs + finishing kernel of the decode path are exercised."""
    from clipcap_amd.engine import DecodeSession
    from clipcap_amd.model.gpt2 import GPT2LM
    torch.manual_seed(3)
    lm = GPT2LM(n_embd=256, n_layer=2, n_head=4, vocab_size=1000, n_positions=32).to("cuda")
    with torch.no_grad():
        for n, p in lm.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.02)
    eng = lm.engine
    x = torch.randn(12, 7, 256, device="cuda") * 0.5
    full = eng.logits(x)
    sess = DecodeSession(eng, 12, 16)
    l0 = sess.forward(x[:, :3]).clone()
    scale = full.abs().max().item()
    assert (l0 - full[:, 2]).abs().max().item() <= 4e-3 * max(1.0, scale)
    for t in range(3, 7):
        lt = sess.forward(x[:, t:t + 1])
        assert (lt - full[:, t]).abs().max().item() <= 4e-3 * max(1.0, scale), t


def test_sampling_decoders_run_on_kv_cache_and_match_first_step_distribution():
    """generate_nucleus_sampling / generate_no_beam / generate: the GPT-2 steps run on the KV-cached HIP path; the pre-sampling
    distribution of the first step equals the reference's (tests/golden/filters.npz, captured from base.py:165-181)."""
    from types import SimpleNamespace
    from clipcap_amd.engine import DecodeSession
    from clipcap_amd.inference import generate_no_beam, generate_nucleus_sampling
    from clipcap_amd.inference.utils import nucleus_distribution
    from clipcap_amd.model.gpt2 import GPT2LM
    f = load_golden("filters")
    g = load_golden("beam_tiny")
    D, n_layer, n_head, V, npos = [int(v) for v in g["cfg"]]
    lm = GPT2LM(n_embd=D, n_layer=n_layer, n_head=n_head, vocab_size=V, n_positions=npos)
    lm.load_state_dict(sd_of(g), strict=False)
    model = SimpleNamespace(language_model=lm.to("cuda"))
    pref = torch.from_numpy(f["nucleus.prefix"]).cuda()
    logits = DecodeSession(lm.engine, 1, 16).forward(pref)
    p = nucleus_distribution(logits, top_p=0.8).cpu().numpy()
    assert np.abs(p - f["nucleus.final_p"]).max() <= 2e-2 and abs(p.sum() - 1.0) <= 1e-5
    assert set(np.nonzero(p[0])[0]) == set(np.nonzero(f["nucleus.final_p"][0])[0]) or np.abs(p - f["nucleus.final_p"]).max() <= 2e-2
    # the product path: cc_sample_step on the same logits gives the same distribution as the torch restatement above
    from clipcap_amd.engine import sample_step
    _, pk = sample_step(logits, torch.rand(1, device="cuda"), top_p=0.8, mode=0, return_probs=True)
    assert np.abs(pk.cpu().numpy() - p).max() <= 2e-6
    assert np.abs(pk.cpu().numpy() - f["nucleus.final_p"]).max() <= 2e-2
    tok = FakeTokenizer(V, 96)
    torch.manual_seed(0)
    a = generate_nucleus_sampling(model, tok, pref, number_to_generate=2, entry_length=6, top_p=0.8)
    torch.manual_seed(0)
    b = generate_nucleus_sampling(model, tok, pref, number_to_generate=2, entry_length=6, top_p=0.8)
    assert a == b and len(a) == 2 and all(isinstance(t, str) for t in a)
    c = generate_no_beam(model, tok, pref, entry_length=5, sweep=False, top_p=0.9)
    assert len(c) == 1 and len(c[0].split()) <= 5
    assert len(generate_no_beam(model, tok, pref, entry_length=2)) == 33          # the reference's 11 x 3 sweep (base.py:229-230)
    # generate() goes through inference/no_beam.py's variant (generate.py:34-41): number_to_generate captions, each starting with
    # the bos/text-prefix tokens, "." (the tokenizer's id for it) never inside a caption
    from clipcap_amd.inference import generate
    lm128 = GPT2LM(n_embd=D, n_layer=n_layer, n_head=n_head, vocab_size=V, n_positions=128).to("cuda")   # 4 + 2*3 + 67 positions
    full = SimpleNamespace(language_model=lm128, transformer_mapper=lambda e: pref)
    caps = generate(full, tok, torch.zeros(1, 8, device="cuda"), number_to_generate=3, text_prefix="ab")
    head = [int(t) for t in tok.encode(tok.bos_token + "ab")]
    dot = tok.encode(".")[0]
    assert len(caps) == 3
    for cap in caps:
        ids = [int(t) for t in cap.split()]
        assert ids[:len(head)] == head and dot not in ids[len(head):] and len(ids) <= len(head) + 67
    # batched prefixes (the reference is batch-1): every row is decoded in the same device loop
    from clipcap_amd.inference.base import sample_tokens
    gen = torch.Generator(device="cuda").manual_seed(5)
    toks, stop_pos = sample_tokens(model, pref.repeat(5, 1, 1), entry_length=7, stop_token=96, mode=0, top_p=0.8, generator=gen)
    assert toks.shape[0] == 5 and toks.shape[1] <= 7 and int(toks.max()) < V and int(toks.min()) >= 0
    assert all(int(stop_pos[r]) == toks.shape[1] or int(toks[r, int(stop_pos[r])]) == 96 for r in range(5))


def test_checkpoint_resume_restores_optimizer_state(tmp_path):
    """Training state written by CheckpointSaver resumes bit-exactly (weights, AdamW moments, step counter) — the reference has
    no resume path; the next step after resume equals the next step of the uninterrupted run."""
    from clipcap_amd.train.callback import CheckpointSaver, resume
    m, g = _model_from_train_fixture()
    m.train()
    tokens, embeds = torch.from_numpy(g["in.tokens"]).cuda(), torch.from_numpy(g["in.embeds"]).cuda()
    for _ in range(2):
        m.fused_step((tokens.clone(), embeds), lr=1e-3)
    CheckpointSaver(str(tmp_path), "r").save_final_checkpoint(m)
    l3 = float(m.fused_step((tokens.clone(), embeds), lr=1e-3))
    after3 = {k: v.clone() for k, v in m.transformer_mapper.state_dict().items()}
    m2, _ = _model_from_train_fixture()
    m2.train()
    resume(m2, str(tmp_path / "r_final.ckpt"))
    assert m2._opt_step == 2
    l3b = float(m2.fused_step((tokens.clone(), embeds), lr=1e-3))
    assert abs(l3 - l3b) <= 1e-6
    for k, v in m2.transformer_mapper.state_dict().items():
        assert torch.allclose(v, after3[k], atol=1e-7, rtol=0), k


def test_in_place_parameter_writes_after_to_device_refresh_the_operand_copies():
    """after .to(device) the parameters are re-pointed with ``p.data = view`` and carry their own version
    counters; load_state_dict / copy_ / a torch optimizer writing through them must still refresh the bf16 operand arena."""
    m, g = _model_from_train_fixture()
    emb = torch.from_numpy(g["in.embeds"]).cuda()
    with torch.no_grad():
        out0 = m.transformer_mapper(emb).clone()
        lg0 = m.language_model(inputs_embeds=torch.randn(1, 5, 64, device="cuda", generator=torch.Generator("cuda").manual_seed(1))).logits.clone()
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        for k in sd:
            if k.endswith("fc1.weight") or k.endswith("c_fc.weight"):
                sd[k] = sd[k] * 1.5
        m.load_state_dict(sd)                                     # in-place writes through the re-pointed parameters
        out1 = m.transformer_mapper(emb).clone()
        lg1 = m.language_model(inputs_embeds=torch.randn(1, 5, 64, device="cuda", generator=torch.Generator("cuda").manual_seed(1))).logits.clone()
        assert (out1 - out0).abs().max().item() > 1e-3 and (lg1 - lg0).abs().max().item() > 1e-4, "stale bf16 weights after load_state_dict"
        # the same weights loaded BEFORE .to(device) (the path the other tests take) give the same outputs
        m2, _ = _model_from_train_fixture()
        m2 = m2.cpu()
        m2.load_state_dict(sd)
        m2 = m2.to("cuda")
        assert torch.equal(m2.transformer_mapper(emb), out1)
        # a plain torch optimizer stepping the re-pointed parameters is seen as well
        p = m.transformer_mapper._arena_params["transformer.layers.0.mlp.fc2.weight"]
        p.mul_(0.5)
        assert (m.transformer_mapper(emb) - out1).abs().max().item() > 1e-4


def test_mapper_autograd_bridge_matches_golden_gradients():
    """``mapper(x)`` under autograd keeps its activations and ``.backward()`` returns the reference's gradients
    (tests/golden/mapper_tiny: loss = out.square().mean()); a graph whose activations were overwritten raises instead of returning
    gradients of the wrong forward."""
    from clipcap_amd.model.mapper import TransformerMapper
    g = load_golden("mapper_tiny")
    E, D, P, L, H, N, B = [int(v) for v in g["dims"]]
    m = TransformerMapper(E, D, L, P, H, N)
    m.load_state_dict(sd_of(g))
    m = m.to("cuda")
    x = torch.from_numpy(g["in.x"]).cuda()
    out = m(x)
    assert out.requires_grad
    out.square().mean().backward()
    for k, p in m.named_parameters():
        ref = torch.from_numpy(g["grad." + k])
        rel = float((p.grad.cpu() - ref).norm() / ref.norm().clamp_min(1e-12))
        assert rel <= 6e-2, (k, rel)
    stale = m(x)
    m(x)                                              # a second saving forward overwrites the workspace of `stale`'s graph
    with pytest.raises(RuntimeError, match="overwritten"):
        stale.sum().backward()
    with torch.no_grad():
        assert not m(x).requires_grad


def test_train_cli_resume_from_restores_schedule_and_state(tmp_path):
    """--resume-from: 2 epochs, then a second process-equivalent call resuming from epoch 0's checkpoint reproduces the
    uninterrupted run's epoch-1 weights (same AdamW moments, optimizer step and LR-schedule position)."""
    import argparse
    from clipcap_amd.model import add_model_args
    from clipcap_amd.model.gpt2 import GPT2LM
    from clipcap_amd.train import add_training_args, train
    _write_dataset(tmp_path / "ds", n=48, E=24, shards=(20, 28))
    GPT2LM(n_embd=64, n_layer=2, n_head=4, vocab_size=157, n_positions=96).save_pretrained(str(tmp_path / "lm"))
    tok = FakeTokenizer()

    def run(out, extra):
        torch.manual_seed(0)
        args = add_model_args(add_training_args(argparse.ArgumentParser())).parse_args([
            "--input-dataset", str(tmp_path / "ds"), "--output-folder", str(tmp_path / out), "--language-model", str(tmp_path / "lm"),
            "--batch-size", "16", "--epochs", "2", "--optimizer-lr", "2e-3", "--scheduler-warmup-steps", "2", "--checkpoint-filename-prefix",
            "t", "--prefix-length", "4", "--projection-length", "4", "--transformer-layers", "2", "--transformer-attention-heads", "4",
            "--logging-frequency", "1000"] + extra)
        assert train(args, tokenizer=tok) == 0
        return torch.load(tmp_path / out / "t_final.ckpt", map_location="cpu")

    a = run("a", [])
    b = run("b", ["--resume-from", str(tmp_path / "a" / "t_epoch_0.ckpt")])
    assert a["optimizer_step"] == b["optimizer_step"] == 6 and a["step"] == b["step"]
    for k, v in a["state_dict"].items():
        assert torch.allclose(v, b["state_dict"][k], atol=1e-6, rtol=0), k
    for part in a["optimizer_state"]:
        assert torch.allclose(a["optimizer_state"][part]["m"], b["optimizer_state"][part]["m"], atol=1e-7, rtol=1e-5)


def test_mapper_forward_with_attention_matches_reference_probabilities():
    """a4: the (b, n, m, h) attention tensor MultiHeadAttention.forward returns (attention.py:32-42), per layer, vs the reference's own
    values (tests/golden/mapper_tiny att.*, mapper_hd96 att.*)."""
    from clipcap_amd.model.mapper import TransformerMapper
    for name in ("mapper_tiny", "mapper_hd96"):
        g = load_golden(name)
        E, D, P, L, H, N, B = [int(v) for v in g["dims"]]
        m = TransformerMapper(E, D, L, P, H, N)
        m.load_state_dict(sd_of(g))
        m = m.to("cuda")
        out, atts = m.forward_with_attention(torch.from_numpy(g["in.x"]).cuda())
        assert len(atts) == N and (out.cpu() - torch.from_numpy(g["out"])).abs().max().item() <= 3e-2
        for i, a in enumerate(atts):
            ref = torch.from_numpy(g[f"att.{i}"])
            assert a.shape == ref.shape
            assert (a.cpu() - ref).abs().max().item() <= 5e-3, (name, i)
            assert (a.sum(dim=2) - 1).abs().max().item() <= 1e-5


def test_language_model_logits_are_differentiable_vs_reference_gradients():
    """``lm(inputs_embeds=x).logits`` under autograd (cc_gpt2_logits_bwd): gradient wrt the inputs and wrt every GPT-2 parameter of
    loss = logits.square().mean() against the reference's own autograd (tests/golden/gpt2_tiny: grad.in.x, grad.*)."""
    from clipcap_amd.model.gpt2 import GPT2LM
    g = load_golden("gpt2_tiny")
    D, n_layer, n_head, V, npos = [int(v) for v in g["cfg"]]
    lm = GPT2LM(n_embd=D, n_layer=n_layer, n_head=n_head, vocab_size=V, n_positions=npos)
    lm.load_state_dict(sd_of(g), strict=False)
    lm = lm.to("cuda")
    x = torch.from_numpy(g["in.x"]).cuda().requires_grad_(True)
    logits = lm(inputs_embeds=x).logits
    assert logits.requires_grad and (logits.detach().cpu() - torch.from_numpy(g["logits"])).abs().max().item() <= 3e-2
    logits.square().mean().backward()
    ref = torch.from_numpy(g["grad.in.x"])
    assert float((x.grad.cpu() - ref).norm() / ref.norm()) <= 3e-2
    n = 0
    for k, p in lm.named_parameters():
        if "grad." + k in g and "lm_head" not in k:
            r = torch.from_numpy(g["grad." + k])
            rel = float((p.grad.cpu() - r).norm() / r.norm().clamp_min(1e-12))
            assert rel <= 6e-2, (k, rel)
            n += 1
    assert n >= 2 + 12 * n_layer + 2
    # a graph whose activations another training-mode pass has overwritten refuses to run backward
    stale = lm(inputs_embeds=x.detach().requires_grad_(True)).logits
    lm(inputs_embeds=x.detach().requires_grad_(True))
    with pytest.raises(RuntimeError, match="backward"):
        stale.sum().backward()
    # inputs only (frozen LM): no parameter gradients are produced, the input gradient is the same
    lm.zero_grad()
    lm.requires_grad_(False)
    x2 = torch.from_numpy(g["in.x"]).cuda().requires_grad_(True)
    lm(inputs_embeds=x2).logits.square().mean().backward()
    assert torch.allclose(x2.grad, x.grad, rtol=1e-4, atol=1e-8) and all(p.grad is None for p in lm.parameters())


def test_model_forward_logits_autograd_equals_fused_training_step():
    """ClipCapModel.forward(...).logits -> the reference's loss expression in torch (model.py:103-109) -> backward(): the mapper
    gradients equal the ones the fused kernel chain (training_step) produces for the same batch."""
    m, g = _model_from_train_fixture("prefix_only")
    m.train()
    tokens, embeds = torch.from_numpy(g["in.tokens"]).cuda(), torch.from_numpy(g["in.embeds"]).cuda()
    L = m.config.prefix_length
    tk = tokens.clamp_min(0)
    logits = m(tk, embeds, tokens.ge(0)).logits
    loss = torch.nn.functional.cross_entropy(logits[:, L - 1:-1].reshape(-1, logits.shape[-1]), tk.flatten(), ignore_index=0)
    loss.backward()
    auto = {k: p.grad.clone() for k, p in m.transformer_mapper.named_parameters()}
    assert abs(float(loss) - float(g["losses"][0])) <= 3e-2
    m.zero_grad()
    eng = m.engine
    eng.zero_grad()
    eng.forward_backward(tokens.clone(), embeds)
    fused = m.transformer_mapper.engine.views(m.transformer_mapper.engine.arena.g32)
    for k, v in auto.items():
        rel = float((v - fused[k]).norm() / fused[k].norm().clamp_min(1e-12))
        assert rel <= 3e-2, (k, rel)
