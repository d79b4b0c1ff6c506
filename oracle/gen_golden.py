"""TEST INFRASTRUCTURE ONLY — generates tests/golden/*.npz by importing the reference (build container only).

Run:  python oracle/gen_golden.py      (needs /root/reference; never runs on the GPU box)

The reference ships no tests / golden vectors, so these fixtures — outputs of the reference's
own PyTorch path on seeded inputs — are what pins the oracle.  Only *data* is written: inputs,
state dicts and the reference's outputs.  ``pytorch_lightning`` is not installed here, so a 6-line stand-in
module is registered before import (LightningModule = nn.Module + no-op save_hyperparameters/log).
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _install_pl_stub():
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    pl.Callback = object
    pl.Trainer = object
    sys.modules["pytorch_lightning"] = pl


def _np(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


class FakeTokenizer:
    """The hot path only needs encode(eos)[0] == stop id and decode(list)->str."""
    eos_token = "<eos>"

    def __init__(self, eos_id):
        self.eos_id = eos_id

    def encode(self, s):
        return [self.eos_id]

    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)


def main():
    _install_pl_stub()
    sys.path.insert(0, REF)
    torch.set_num_threads(4)
    from transformers import GPT2Config, GPT2LMHeadModel
    from clipcap.model.mapper import TransformerMapper, TransformerMapperWindowed
    from clipcap.model.model import ClipCapModel, ClipCapModelPrefixOnly
    from clipcap.model.config import Config, TrainingConfig
    from clipcap.encoders.config import EncoderConfig
    from clipcap.inference import base as ibase
    os.makedirs(OUT, exist_ok=True)

    # ---------------- (1) mapper: tiny + shape-faithful (1 layer) ----------------
    for name, (E, D, P, L, H, N, B) in {"mapper_tiny": (32, 64, 4, 4, 4, 2, 3),
                                        "mapper_faithful": (64, 192, 10, 10, 2, 2, 2)}.items():
        torch.manual_seed(101)
        m = TransformerMapper(E, D, L, P, H, N)
        x = torch.randn(B, E)
        out = m(x)
        # per-layer activations + attention probs
        lin = m.linear(x).view(B, P, -1)
        h = torch.cat((lin, m.prefix_const.unsqueeze(0).expand(B, L, D)), dim=1)
        _, atts = m.transformer.forward_with_attention(h)
        loss = out.square().mean()
        loss.backward()
        d = {"in.x": x.numpy(), "out": out.detach().numpy(), "loss": loss.detach().numpy(),
             "dims": np.array([E, D, P, L, H, N, B])}
        for i, a in enumerate(atts):
            d[f"att.{i}"] = a.detach().numpy()
        for k, v in m.state_dict().items():
            d["sd." + k] = v.numpy()
        for k, v in m.named_parameters():
            d["grad." + k] = v.grad.numpy()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)

    # windowed mapper
    torch.manual_seed(102)
    E, D, P, L, H, N, B, W = 16, 32, 2, 3, 4, 2, 2, 3
    m = TransformerMapperWindowed(E, D, L, P, W, True, H, N)
    x = torch.randn(B, W, E)
    out = m(x)
    d = {"in.x": x.numpy(), "out": out.detach().numpy(), "dims": np.array([E, D, P, L, H, N, B, W])}
    for k, v in m.state_dict().items():
        d["sd." + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "mapper_windowed.npz"), **d)

    # ---------------- (2) GPT-2 tiny ----------------
    torch.manual_seed(103)
    gcfg = GPT2Config(n_embd=64, n_layer=2, n_head=4, vocab_size=211, n_positions=64,
                      resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
    lm = GPT2LMHeadModel(gcfg).eval()
    with torch.no_grad():  # default init leaves biases 0 / LN at identity: perturb so they are exercised
        for n_, p_ in lm.named_parameters():
            if n_.endswith(".bias") or "ln_" in n_:
                p_.add_(0.05 * torch.randn_like(p_))
    xe = (0.5 * torch.randn(2, 12, 64)).requires_grad_(True)
    logits = lm(inputs_embeds=xe).logits
    (logits.square().mean()).backward()
    d = {"in.x": xe.detach().numpy(), "logits": logits.detach().numpy(), "grad.in.x": xe.grad.numpy(),
         "cfg": np.array([64, 2, 4, 211, 64])}
    for k, v in lm.state_dict().items():
        d["sd." + k] = v.numpy()
    for k, v in lm.named_parameters():
        d["grad." + k] = v.grad.numpy()
    mask = torch.ones(2, 12, dtype=torch.bool)
    mask[0, 9:] = False
    mask[1, 11:] = False
    with torch.no_grad():
        d["logits_masked"] = lm(inputs_embeds=xe.detach(), attention_mask=mask).logits.numpy()
    d["mask"] = mask.numpy()
    np.savez_compressed(os.path.join(OUT, "gpt2_tiny.npz"), **d)

    # ---------------- (3) training_step + 3 AdamW/scheduler steps ----------------
    tmp = tempfile.mkdtemp()
    torch.manual_seed(104)
    gcfg = GPT2Config(n_embd=64, n_layer=2, n_head=4, vocab_size=157, n_positions=40,
                      resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
    lm0 = GPT2LMHeadModel(gcfg)
    with torch.no_grad():
        for n_, p_ in lm0.named_parameters():
            if n_.endswith(".bias") or "ln_" in n_:
                p_.add_(0.05 * torch.randn_like(p_))
    lm0.save_pretrained(tmp)
    for mode in ("prefix_only", "full"):
        torch.manual_seed(105)
        enc = EncoderConfig(encoder_embedding_size=24)
        cfg = Config(language_model=tmp, train_language_model=(mode == "full"), prefix_length=3,
                     projection_length=2, transformer_layers=2, transformer_attention_heads=4,
                     encoder_config=enc,
                     training_config=TrainingConfig(optimizer_lr=1e-3, use_deepspeed_optimisers=False,
                                                    scheduler_warmup_steps=2, total_steps=6))
        cls = ClipCapModel if mode == "full" else ClipCapModelPrefixOnly
        model = cls(cfg)
        model.train()
        Bt, cap = 4, 8
        tokens = torch.randint(1, 157, (Bt, cap))
        tokens[0, 5:] = -1
        tokens[2, 7:] = -1
        tokens[1, 2] = 0          # explicit token id 0 (ignored by the loss as a side effect, model.py:109)
        embeds = torch.randn(Bt, 24)
        d = {"in.tokens": tokens.numpy().copy(), "in.embeds": embeds.numpy(),
             "cfg": np.array([24, 64, 2, 3, 4, 2, 4, 2, 157, 40])}  # E D P L H N n_head n_layer V n_pos
        for k, v in model.state_dict().items():
            d["sd." + k] = v.numpy().copy()
        oc = model.configure_optimizers()
        opt, sch = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
        losses, lrs = [], []
        for step in range(3):
            opt.zero_grad()
            loss = model.training_step((tokens.clone(), embeds), 0)
            loss.backward()
            if step == 0:
                with torch.no_grad():
                    d["logits0"] = model(torch.where(tokens < 0, 0, tokens), embeds, tokens.ge(0)).logits.numpy()
                for k, v in model.named_parameters():
                    if v.grad is not None:
                        d["grad0." + k] = v.grad.numpy().copy()
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
            losses.append(float(loss))
            for k, v in model.state_dict().items():
                if mode == "full" or k.startswith("transformer_mapper."):
                    d[f"sd_after{step + 1}." + k] = v.numpy().copy()
        d["losses"] = np.array(losses)
        d["lrs"] = np.array(lrs)
        np.savez_compressed(os.path.join(OUT, f"train_{mode}.npz"), **d)

    # ---------------- (4) generate_beam traces ----------------
    torch.manual_seed(106)
    V = 97
    gcfg = GPT2Config(n_embd=32, n_layer=2, n_head=4, vocab_size=V, n_positions=48,
                      resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
    lmb = GPT2LMHeadModel(gcfg).eval()
    with torch.no_grad():
        for n_, p_ in lmb.named_parameters():
            p_.mul_(6.0 if "wte" in n_ else 1.5)   # peaky distributions so beams hit EOS at varied times
            if n_.endswith(".bias") or "ln_" in n_:
                p_.add_(0.05 * torch.randn_like(p_))

    # make token 50 overwhelmingly likely from position 8 on, so that beam sets using eos=50 all stop early
    # (and one beam stops at step 1 and survives as a frozen beam) — exercises base.py:96-97,119-121
    with torch.no_grad():
        torch.manual_seed(7)
        u = torch.randn(32)
        lmb.transformer.wpe.weight[8:] += 3.0 * u
        lmb.transformer.wpe.weight[7] += 1.5 * u
        lmb.transformer.wte.weight[50] = 0.6 * ((u - u.mean()) / u.std()) * lmb.transformer.ln_f.weight.sign()

    class _M:  # minimal object with the attributes generate_beam touches (base.py:76,81,117)
        language_model = lmb

    d = {"cfg": np.array([32, 2, 4, V, 48])}
    for k, v in lmb.state_dict().items():
        d["sd." + k] = v.numpy()
    cases = []
    for seed, eos, entry in [(1, 50, 12), (2, 50, 12), (3, 50, 10), (4, 40, 16), (5, 7, 9)]:
        torch.manual_seed(1000 + seed)
        pref = torch.randn(1, 4, 32)
        tok = FakeTokenizer(eos)
        texts = ibase.generate_beam(_M, tok, pref, beam_size=5, entry_length=entry, temperature=1.0)
        d[f"beam{seed}.prefix"] = pref.numpy()
        d[f"beam{seed}.best"] = np.array([int(s) for s in texts[0].split()] if texts[0] else [], dtype=np.int64)
        d[f"beam{seed}.meta"] = np.array([eos, entry, 5])
        cases.append(seed)
    d["cases"] = np.array(cases)
    # temperature != 1 case
    torch.manual_seed(2001)
    pref = torch.randn(1, 4, 32)
    texts = ibase.generate_beam(_M, FakeTokenizer(5), pref, beam_size=3, entry_length=10, temperature=0.7)
    d["beamT.prefix"] = pref.numpy()
    d["beamT.best"] = np.array([int(s) for s in texts[0].split()] if texts[0] else [], dtype=np.int64)
    d["beamT.meta"] = np.array([5, 10, 3])
    np.savez_compressed(os.path.join(OUT, "beam_tiny.npz"), **d)

    # ---------------- (5) filters ----------------
    torch.manual_seed(107)
    lg = torch.randn(50) * 3
    d = {"in.logits": lg.numpy()}
    d["topk5"] = ibase.top_k_top_p_filtering(lg.clone(), top_k=5).numpy()
    d["topp08"] = ibase.top_k_top_p_filtering(lg.clone(), top_p=0.8).numpy()
    d["topk10_topp05"] = ibase.top_k_top_p_filtering(lg.clone(), top_k=10, top_p=0.5).numpy()
    toks = torch.tensor([3, 7, 7, 20])
    d["in.tokens"] = toks.numpy()
    d["rep12"] = ibase.repetition_penalty_apply(lg.clone(), toks, 1.2).numpy()
    lg2 = lg.clone()
    lg2[7] = 11.0  # a logit whose VALUE equals stop_token=11 -> exercises the reference's value comparison
    d["in.logits2"] = lg2.numpy()
    d["lenpen"] = ibase.sentence_length_penalty_apply(lg2.clone(), toks, 11, 4, 50, 1.0).numpy()
    # nucleus final_p: re-run the reference's arithmetic via its function is not separable (it samples);
    # capture it by calling with a patched multinomial.
    captured = {}
    real_multinomial = torch.multinomial

    def fake_multinomial(pr, num_samples=1, **kw):
        captured.setdefault("p", pr.clone())
        return pr.argmax(dim=-1, keepdim=True)

    torch.multinomial = fake_multinomial
    try:
        torch.manual_seed(108)
        pref = torch.randn(1, 4, 32)
        ibase.generate_nucleus_sampling(_M, FakeTokenizer(96), pref, entry_length=2, top_p=0.8)
    finally:
        torch.multinomial = real_multinomial
    d["nucleus.prefix"] = pref.numpy()
    d["nucleus.final_p"] = captured["p"].numpy()
    np.savez_compressed(os.path.join(OUT, "filters.npz"), **d)

    # ---------------- (6) autocast-drift yardstick (numbers only) ----------------
    torch.manual_seed(109)
    m = TransformerMapper(512, 768, 10, 10, 8, 8)
    x = torch.randn(4, 512)
    with torch.no_grad():
        ref = m(x)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            o16 = m(x).float()
    np.savez_compressed(os.path.join(OUT, "drift.npz"),
                        mapper_bf16_autocast_maxabs=np.array(float((ref - o16).abs().max())),
                        mapper_out_absmax=np.array(float(ref.abs().max())))
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
