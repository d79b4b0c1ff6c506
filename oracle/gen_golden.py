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


def _sampled(d, key, t):
    """Stores a large tensor as (norm, evenly strided sample) — tests/seeded.py:sample_idx gives the positions."""
    from tests.seeded import sample_idx
    a = t.detach().numpy().reshape(-1)
    d[key + ".norm"] = np.array(float(np.sqrt((a.astype(np.float64) ** 2).sum())))
    d[key + ".sample"] = a[sample_idx(a.size)].copy()


def _seeded_clipcap(tmp, *, E, D, P, L, H, N, NL, n_head, V, NPOS, seed, full):
    """The REFERENCE's ClipCapModel(PrefixOnly) carrying tests/seeded.py parameters (GPT-2 dropout 0 so train mode is deterministic)."""
    from transformers import GPT2Config, GPT2LMHeadModel
    from clipcap.model.model import ClipCapModel, ClipCapModelPrefixOnly
    from clipcap.model.config import Config, TrainingConfig
    from clipcap.encoders.config import EncoderConfig
    from tests import seeded
    gsd = seeded.state_dict(seeded.gpt2_shapes(D, NL, V, NPOS), seed)
    msd = seeded.state_dict(seeded.mapper_shapes(E, D, P, L, N), seed + 1)
    lm = GPT2LMHeadModel(GPT2Config(n_embd=D, n_layer=NL, n_head=n_head, vocab_size=V, n_positions=NPOS, resid_pdrop=0.0, embd_pdrop=0.0,
                                    attn_pdrop=0.0))
    missing = lm.load_state_dict({k: torch.from_numpy(v) for k, v in gsd.items()}, strict=False)
    assert all("attn.bias" in k or "masked_bias" in k or k == "lm_head.weight" for k in missing.missing_keys), missing
    lm.save_pretrained(tmp)
    cfg = Config(language_model=tmp, train_language_model=full, prefix_length=L, projection_length=P, transformer_layers=N,
                 transformer_attention_heads=H, encoder_config=EncoderConfig(encoder_embedding_size=E),
                 training_config=TrainingConfig(optimizer_lr=1e-3, use_deepspeed_optimisers=False, scheduler_warmup_steps=2, total_steps=6))
    model = (ClipCapModel if full else ClipCapModelPrefixOnly)(cfg)
    model.transformer_mapper.load_state_dict({k: torch.from_numpy(v) for k, v in msd.items()})
    model.train()
    return model, gsd, msd


def _full_model_fixture(name, *, E, D, P, L, H, N, NL, n_head, V, NPOS, seed, full, B=2, cap=40):
    """Shape-faithful whole-model fixture (BASELINE configs[1] / configs[3] architecture at B=2): the reference's logits (sub-sampled),
    loss and gradients (sub-sampled) for seeded parameters that the tests regenerate instead of loading."""
    from tests import seeded
    tmp = tempfile.mkdtemp()
    model, gsd, msd = _seeded_clipcap(tmp, E=E, D=D, P=P, L=L, H=H, N=N, NL=NL, n_head=n_head, V=V, NPOS=NPOS, seed=seed, full=full)
    gen = torch.Generator().manual_seed(seed + 7)
    tokens = torch.randint(1, V, (B, cap), generator=gen)
    tokens[0, cap - 6:] = -1
    tokens[1, 3] = 0
    embeds = torch.randn(B, E, generator=gen)
    d = {"in.tokens": tokens.numpy().copy(), "in.embeds": embeds.numpy(),
         "cfg": np.array([E, D, P, L, H, N, n_head, NL, V, NPOS, seed, int(full)]),
         "param_checksum": np.concatenate([seeded.checksum(gsd), seeded.checksum(msd)])}
    loss = model.training_step((tokens.clone(), embeds), 0)
    loss.backward()
    d["loss"] = np.array(float(loss))
    for k, v in model.named_parameters():
        if v.grad is not None and (full or k.startswith("transformer_mapper.")):
            _sampled(d, "grad0." + k, v.grad)
    with torch.no_grad():
        model.eval()
        logits = model(torch.where(tokens < 0, 0, tokens), embeds, tokens.ge(0)).logits          # (B, T, V)
        prefix = model.transformer_mapper(embeds)
    cols = seeded.sample_idx(V, 1024)
    d["logits.cols"] = logits[:, :, cols].numpy()                       # every row, 1024 strided vocabulary columns
    d["logits.rows"] = logits[:, [L - 1, L + 7, L + cap - 2], :].numpy()  # three full rows per sample
    d["logits.absmax"] = np.array(float(logits.abs().max()))
    d["prefix"] = prefix.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, "loss", float(loss), "logits absmax", float(logits.abs().max()))


def main():
    _install_pl_stub()
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    only = set(sys.argv[1:])
    if only:
        return _main_new(only)
    _main_round1()
    _main_new(set())


def _main_new(only):
    """Fixtures added in round 2 (selectable by name on the command line; no name = all)."""
    torch.set_num_threads(8)
    from tests import seeded
    from clipcap.model.mapper import TransformerMapper
    from clipcap.inference import no_beam as inb, nucleus_sampling as ins
    from transformers import GPT2Config, GPT2LMHeadModel
    os.makedirs(OUT, exist_ok=True)

    def want(n):
        return not only or n in only

    # ---- (1b) the shape-faithful mapper of SURVEY.md 8c: E=512, D=768, P=L=10, H=8 (hd 96), N=1, B=2, seeded parameters ----
    if want("mapper_faithful"):
        E, D, P, L, H, N, B, seed = 512, 768, 10, 10, 8, 1, 2, 4101
        msd = seeded.state_dict(seeded.mapper_shapes(E, D, P, L, N), seed)
        m = TransformerMapper(E, D, L, P, H, N)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in msd.items()})
        x = torch.randn(B, E, generator=torch.Generator().manual_seed(seed))
        out = m(x)
        lin = m.linear(x).view(B, P, -1)
        h = torch.cat((lin, m.prefix_const.unsqueeze(0).expand(B, L, D)), dim=1)
        _, atts = m.transformer.forward_with_attention(h)
        loss = out.square().mean()
        loss.backward()
        d = {"in.x": x.numpy(), "out": out.detach().numpy(), "loss": loss.detach().numpy(), "dims": np.array([E, D, P, L, H, N, B, seed]),
             "param_checksum": seeded.checksum(msd), "att.0": atts[0].detach().numpy()}
        for k, v in m.named_parameters():
            _sampled(d, "grad." + k, v.grad)
        np.savez_compressed(os.path.join(OUT, "mapper_faithful.npz"), **d)

    # ---- (7) the sampling variants generate() uses: no_beam.py:10-82 (stop on ".", repetition penalty 1.2) and
    #          nucleus_sampling.py:9-74; torch.multinomial patched to record the pre-sampling distribution of every step and to
    #          force a deterministic token (rank step % 3 of the distribution) so that the history has repeats ----
    if want("sampling_steps"):
        WTE_SCALE = 5.0
        b = np.load(os.path.join(OUT, "gpt2_tiny.npz"))         # the D=64 / V=211 GPT-2 of fixture (2), wte scaled for peakier steps
        D, n_layer, n_head, V, npos = [int(v) for v in b["cfg"]]
        lmb = GPT2LMHeadModel(GPT2Config(n_embd=D, n_layer=n_layer, n_head=n_head, vocab_size=V, n_positions=npos, resid_pdrop=0.0,
                                         embd_pdrop=0.0, attn_pdrop=0.0)).eval()
        lmb.load_state_dict({k[3:]: torch.from_numpy(b[k]) for k in b.files if k.startswith("sd.")}, strict=False)
        with torch.no_grad():
            lmb.transformer.wte.weight.mul_(WTE_SCALE)

        class _M:
            language_model = lmb

        class DotTokenizer(FakeTokenizer):
            def encode(self, s):
                return [self.eos_id]            # "." -> the stop id under test

        real_multinomial = torch.multinomial
        d = {"wte_scale": np.array(WTE_SCALE)}
        for case, (fn, kw, stop, head) in {
                "no_beam": (inb.generate_no_beam, dict(number_to_generate=1, top_p=0.9, top_k=0.0, temperature=0.9, repetition_penalty=1.2), 200,
                            torch.tensor([[5, 17, 5]])),
                "no_beam_topk": (inb.generate_no_beam, dict(number_to_generate=1, top_p=0.7, top_k=12, temperature=1.0, repetition_penalty=1.5), 200, None),
                "nucleus": (ins.generate_nucleus_sampling, dict(number_to_generate=1, top_p=0.8, top_k=0, temperature=1.0), 200, None)}.items():
            rec, forced = [], []

            def fake_multinomial(pr, num_samples=1, **kw_):
                pr2 = pr.reshape(-1, pr.shape[-1])
                rec.append(pr2[0].clone())
                k = len(rec) % 3
                nz = int((pr2[0] > 0).sum())
                tok = pr2[0].argsort(descending=True, stable=True)[min(k, nz - 1)].reshape(1)
                if tok.item() == stop:          # never force the stop token: keep the trace at full length
                    tok = pr2[0].argsort(descending=True, stable=True)[0].reshape(1) if nz == 1 else \
                        pr2[0].argsort(descending=True, stable=True)[(min(k, nz - 1) + 1) % nz].reshape(1)
                forced.append(int(tok))
                return tok.reshape(pr.shape[:-1] + (1,)) if pr.dim() > 1 else tok

            torch.multinomial = fake_multinomial
            try:
                torch.manual_seed(3000 + len(d))
                pref = torch.randn(1, 4, D)
                text = fn(_M, DotTokenizer(stop), pref, text_prefix_tokens=head, entry_length=10, **kw)
            finally:
                torch.multinomial = real_multinomial
            d[case + ".prefix"] = pref.numpy()
            d[case + ".head"] = (head if head is not None else torch.zeros(1, 0, dtype=torch.int64)).numpy()
            d[case + ".probs"] = torch.stack(rec).numpy()
            d[case + ".forced"] = np.array(forced, dtype=np.int64)
            d[case + ".text"] = np.array([int(s) for s in text[0].split()], dtype=np.int64)
            d[case + ".kw"] = np.array([kw["top_p"], float(kw["top_k"]), kw["temperature"], kw.get("repetition_penalty", 1.0), stop])
        np.savez_compressed(os.path.join(OUT, "sampling_steps.npz"), **d)

    # ---- (8) BASELINE configs[1] architecture, full depth: 8-layer mapper + 12-layer GPT-2-small, frozen LM ----
    if want("config2_full"):
        _full_model_fixture("config2_full", E=512, D=768, P=10, L=10, H=8, N=8, NL=12, n_head=12, V=50257, NPOS=1024, seed=4201, full=False)
    # ---- (9) BASELINE configs[3] architecture, full depth: E=1024 -> D=1024 mapper (hd 128) + 24-layer GPT-2-medium, full finetune ----
    if want("config4_full"):
        _full_model_fixture("config4_full", E=1024, D=1024, P=10, L=10, H=8, N=8, NL=24, n_head=16, V=50257, NPOS=1024, seed=4301, full=True)

    # ---- (10) BASELINE configs[4]: beam-5 decode at GPT-2-medium width (D=1024, 16 heads, V=50257), 4 of 24 layers, peaky wte ----
    if want("beam_medium"):
        from clipcap.inference import base as ibase
        D, NL, n_head, V, NPOS, seed = 1024, 4, 16, 50257, 128, 4401
        gsd = seeded.state_dict(seeded.gpt2_shapes(D, NL, V, NPOS), seed)
        gsd["transformer.wte.weight"] = gsd["transformer.wte.weight"] * 2.0          # peakier next-token distributions (top-beam margins ~0.1)
        lm = GPT2LMHeadModel(GPT2Config(n_embd=D, n_layer=NL, n_head=n_head, vocab_size=V, n_positions=NPOS, resid_pdrop=0.0, embd_pdrop=0.0,
                                        attn_pdrop=0.0)).eval()
        lm.load_state_dict({k: torch.from_numpy(v) for k, v in gsd.items()}, strict=False)

        class _M2:
            language_model = lm

        d = {"cfg": np.array([D, NL, n_head, V, NPOS, seed]), "wte_scale": np.array(2.0), "param_checksum": seeded.checksum(gsd)}
        gen = torch.Generator().manual_seed(seed)
        for i in range(2):
            pref = torch.randn(1, 10, D, generator=gen) * 0.5
            # run A: a stop token that is never produced; run B: stop = the 5th token of run A's caption, so beams freeze mid-way
            texts = ibase.generate_beam(_M2, FakeTokenizer(V - 1), pref, beam_size=5, entry_length=10, temperature=1.0)
            best = [int(s) for s in texts[0].split()]
            d[f"beam{i}a.prefix"] = pref.numpy()
            d[f"beam{i}a.best"] = np.array(best, dtype=np.int64)
            d[f"beam{i}a.meta"] = np.array([V - 1, 10, 5])
            eos = best[4]
            texts = ibase.generate_beam(_M2, FakeTokenizer(eos), pref, beam_size=5, entry_length=10, temperature=1.0)
            d[f"beam{i}b.prefix"] = pref.numpy()
            d[f"beam{i}b.best"] = np.array([int(s) for s in texts[0].split()] if texts[0] else [], dtype=np.int64)
            d[f"beam{i}b.meta"] = np.array([eos, 10, 5])
        np.savez_compressed(os.path.join(OUT, "beam_medium.npz"), **d)

    # ---- (11) round 4 — BASELINE configs[2] architecture at full depth: the configs[1] mapper + 12-layer GPT-2-small, FULL finetune
    #           (the LM's weight gradients at GPT-2-small shapes; config4_full covers them at GPT-2-medium shapes only) ----
    if want("config3_full"):
        _full_model_fixture("config3_full", E=512, D=768, P=10, L=10, H=8, N=8, NL=12, n_head=12, V=50257, NPOS=1024, seed=4501, full=True)

    # ---- (12) round 4 — BASELINE configs[4] at GPT-2-medium DEPTH: the reference's generate_beam (inference/base.py:55-132) on the seeded
    #           24-layer model, 2 prefixes, entry_length 12, each also with a stop token that freezes beams mid-way ----
    if want("beam_deep"):
        from clipcap.inference import base as ibase
        D, NL, n_head, V, NPOS, seed, EL = 1024, 24, 16, 50257, 128, 4601, 12
        gsd = seeded.state_dict(seeded.gpt2_shapes(D, NL, V, NPOS), seed)
        gsd["transformer.wte.weight"] = gsd["transformer.wte.weight"] * 2.0
        lm = GPT2LMHeadModel(GPT2Config(n_embd=D, n_layer=NL, n_head=n_head, vocab_size=V, n_positions=NPOS, resid_pdrop=0.0, embd_pdrop=0.0,
                                        attn_pdrop=0.0)).eval()
        lm.load_state_dict({k: torch.from_numpy(v) for k, v in gsd.items()}, strict=False)

        class _M3:
            language_model = lm

        d = {"cfg": np.array([D, NL, n_head, V, NPOS, seed]), "wte_scale": np.array(2.0), "param_checksum": seeded.checksum(gsd)}
        gen = torch.Generator().manual_seed(seed)
        for i in range(2):
            pref = torch.randn(1, 10, D, generator=gen) * 0.5
            texts = ibase.generate_beam(_M3, FakeTokenizer(V - 1), pref, beam_size=5, entry_length=EL, temperature=1.0)
            best = [int(s) for s in texts[0].split()]
            d[f"beam{i}a.prefix"] = pref.numpy()
            d[f"beam{i}a.best"] = np.array(best, dtype=np.int64)
            d[f"beam{i}a.meta"] = np.array([V - 1, EL, 5])
            eos = best[5]
            texts = ibase.generate_beam(_M3, FakeTokenizer(eos), pref, beam_size=5, entry_length=EL, temperature=1.0)
            d[f"beam{i}b.prefix"] = pref.numpy()
            d[f"beam{i}b.best"] = np.array([int(s) for s in texts[0].split()] if texts[0] else [], dtype=np.int64)
            d[f"beam{i}b.meta"] = np.array([eos, EL, 5])
            print("beam_deep", i, best, d[f"beam{i}b.best"])
        np.savez_compressed(os.path.join(OUT, "beam_deep.npz"), **d)

    # ---- (14) round 5 — number_to_generate > 1 (inference/base.py:79-130): the entry_length loop re-entered with LIVE state.  GPT-2-medium width,
    #           4 layers (the beam_medium model), 3 generations of 6 tokens; run a: the stop token never appears (the beams keep growing:
    #           6 / 12 / 18 tokens), run b: stop = the 3rd token of run a's first text (beams freeze; later rounds take one step and re-divide) ----
    if want("beam_multi"):
        from clipcap.inference import base as ibase
        D, NL, n_head, V, NPOS, seed, EL, NG = 1024, 4, 16, 50257, 128, 4401, 6, 3
        gsd = seeded.state_dict(seeded.gpt2_shapes(D, NL, V, NPOS), seed)
        gsd["transformer.wte.weight"] = gsd["transformer.wte.weight"] * 2.0
        lm = GPT2LMHeadModel(GPT2Config(n_embd=D, n_layer=NL, n_head=n_head, vocab_size=V, n_positions=NPOS, resid_pdrop=0.0, embd_pdrop=0.0,
                                        attn_pdrop=0.0)).eval()
        lm.load_state_dict({k: torch.from_numpy(v) for k, v in gsd.items()}, strict=False)

        class _M5:
            language_model = lm

        d = {"cfg": np.array([D, NL, n_head, V, NPOS, seed]), "wte_scale": np.array(2.0), "param_checksum": seeded.checksum(gsd)}
        gen = torch.Generator().manual_seed(seed + 7)
        for i in range(2):
            pref = torch.randn(1, 10, D, generator=gen) * 0.5
            texts = ibase.generate_beam(_M5, FakeTokenizer(V - 1), pref, number_to_generate=NG, beam_size=5, entry_length=EL, temperature=1.0)
            assert len(texts) == NG
            d[f"multi{i}a.prefix"] = pref.numpy()
            d[f"multi{i}a.meta"] = np.array([V - 1, EL, 5, NG])
            for r, t in enumerate(texts):
                d[f"multi{i}a.gen{r}"] = np.array([int(x) for x in t.split()], dtype=np.int64)
            eos = int(texts[0].split()[2])
            texts_b = ibase.generate_beam(_M5, FakeTokenizer(eos), pref, number_to_generate=NG, beam_size=5, entry_length=EL, temperature=1.0)
            d[f"multi{i}b.prefix"] = pref.numpy()
            d[f"multi{i}b.meta"] = np.array([eos, EL, 5, NG])
            for r, t in enumerate(texts_b):
                d[f"multi{i}b.gen{r}"] = np.array([int(x) for x in t.split()] if t else [], dtype=np.int64)
            # run c: stop = the most frequent token of run a's last text — every beam runs into it, so the later rounds start with all beams stopped
            last = [int(x) for x in texts[-1].split()]
            eos_c = max(set(last), key=last.count)
            texts_c = ibase.generate_beam(_M5, FakeTokenizer(eos_c), pref, number_to_generate=NG, beam_size=5, entry_length=EL, temperature=1.0)
            d[f"multi{i}c.prefix"] = pref.numpy()
            d[f"multi{i}c.meta"] = np.array([eos_c, EL, 5, NG])
            for r, t in enumerate(texts_c):
                d[f"multi{i}c.gen{r}"] = np.array([int(x) for x in t.split()] if t else [], dtype=np.int64)
            print("beam_multi", i, [len(t.split()) for t in texts], [t for t in texts_b], "c:", eos_c, [t for t in texts_c])
        np.savez_compressed(os.path.join(OUT, "beam_multi.npz"), **d)

    # ---- (15) round 5 — beam RANKING under competition at full depth.  beam_deep's captions are degenerate (a random-init GPT-2 with tied,
    #           scaled-up wte repeats one token: the hidden state is dominated by the last input embedding), so most of its steps are not
    #           close calls.  Here the position embeddings are scaled x8 and wte x0.5, temperature 1.5: every step's winner changes with the
    #           position and the captions do not repeat (asserted: >= 8 distinct tokens of 12).  24-layer GPT-2-medium, 2 prefixes x {no stop, stop mid-way} ----
    if want("beam_varied"):
        from clipcap.inference import base as ibase
        D, NL, n_head, V, NPOS, seed, EL, TEMP = 1024, 24, 16, 50257, 128, 4801, 12, 1.5
        gsd = seeded.state_dict(seeded.gpt2_shapes(D, NL, V, NPOS), seed)
        gsd["transformer.wte.weight"] = gsd["transformer.wte.weight"] * 0.5
        gsd["transformer.wpe.weight"] = gsd["transformer.wpe.weight"] * 8.0
        lm = GPT2LMHeadModel(GPT2Config(n_embd=D, n_layer=NL, n_head=n_head, vocab_size=V, n_positions=NPOS, resid_pdrop=0.0, embd_pdrop=0.0,
                                        attn_pdrop=0.0)).eval()
        lm.load_state_dict({k: torch.from_numpy(v) for k, v in gsd.items()}, strict=False)

        class _M6:
            language_model = lm

        d = {"cfg": np.array([D, NL, n_head, V, NPOS, seed]), "wte_scale": np.array(0.5), "wpe_scale": np.array(8.0), "temperature": np.array(TEMP),
             "param_checksum": seeded.checksum(gsd)}
        gen = torch.Generator().manual_seed(seed)
        for i in range(2):
            pref = torch.randn(1, 10, D, generator=gen) * 0.5
            texts = ibase.generate_beam(_M6, FakeTokenizer(V - 1), pref, beam_size=5, entry_length=EL, temperature=TEMP)
            best = [int(x) for x in texts[0].split()]
            assert len(set(best)) >= 8, best
            d[f"beam{i}a.prefix"] = pref.numpy()
            d[f"beam{i}a.best"] = np.array(best, dtype=np.int64)
            d[f"beam{i}a.meta"] = np.array([V - 1, EL, 5])
            eos = best[5]
            texts = ibase.generate_beam(_M6, FakeTokenizer(eos), pref, beam_size=5, entry_length=EL, temperature=TEMP)
            d[f"beam{i}b.prefix"] = pref.numpy()
            d[f"beam{i}b.best"] = np.array([int(x) for x in texts[0].split()] if texts[0] else [], dtype=np.int64)
            d[f"beam{i}b.meta"] = np.array([eos, EL, 5])
            print("beam_varied", i, best, d[f"beam{i}b.best"])
        np.savez_compressed(os.path.join(OUT, "beam_varied.npz"), **d)

    # ---- (13) round 4 — the sentence-length penalty of no_beam.py:55-60 on rows where it FIRES: utils.py:40-51 multiplies the logits of
    #           history tokens whose VALUE equals the stop-token id, so the rows below carry history tokens whose filtered logit is exactly
    #           float(stop).  The per-step rule is the reference's own call sequence (no_beam.py:45-63) on 1-D logits. ----
    if want("length_penalty"):
        from clipcap.inference import utils as iu
        import torch.nn.functional as F
        d = {}
        gen = torch.Generator().manual_seed(4701)
        cases = [  # V, stop, temperature, repetition penalty, top_p, top_k, desired length, factor, history length
            (97, 13, 1.0, 1.0, 0.9, 0, 50, 1.0, 6), (1000, 13, 0.5, 1.0, 0.95, 0, 10, 3.0, 9), (1000, 7, 1.0, 1.25, 1.0, 40, 4, 1.0, 12),
            (50257, 13, 1.0, 1.0, 0.9, 0, 50, 1.0, 30), (50257, 13, 1.0, 1.0, 0.8, 0, 5, 2.0, 20), (211, 5, 2.0, 1.0, 0.0, 8, 50, 0.0, 3)]
        for ci, (V, stop, temp, rep, top_p, top_k, want_len, factor, hl) in enumerate(cases):
            lg = torch.randn(V, generator=gen) * 3.0
            hist = torch.randint(0, V, (hl,), generator=gen)
            hist[1] = hist[0]                                        # a repeated history token
            # two history tokens and one non-history token whose value after repetition penalty and temperature is exactly float(stop)
            nonh = int([t for t in range(V) if t not in set(hist.tolist())][3])
            for t in (int(hist[0]), int(hist[2]), nonh):
                lg[t] = float(stop) * temp * (rep if t != nonh else 1.0)
            x = lg.clone()
            if rep != 1.0:
                x = iu.repetition_penalty_apply(x, hist, rep)
            x = x / (temp if temp > 0 else 1.0)
            x = iu.top_k_top_p_filtering(x, top_p=top_p, top_k=top_k)
            fired = int((x[hist] == stop).sum())
            x = iu.sentence_length_penalty_apply(x, hist, stop, hist.numel(), want_len, factor)
            pr = F.softmax(x, dim=-1)
            d[f"c{ci}.logits"] = lg.numpy()
            d[f"c{ci}.hist"] = hist.numpy()
            d[f"c{ci}.kw"] = np.array([stop, temp, rep, top_p, top_k, want_len, factor, fired], dtype=np.float64)
            nzi = torch.nonzero(pr > 0).flatten()
            d[f"c{ci}.idx"] = nzi.numpy()
            d[f"c{ci}.probs"] = pr[nzi].numpy()
            print("length_penalty", ci, "history entries at the stop value:", fired, "kept:", nzi.numel())
        np.savez_compressed(os.path.join(OUT, "length_penalty.npz"), **d)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


def _main_round1():
    torch.set_num_threads(4)
    from transformers import GPT2Config, GPT2LMHeadModel
    from clipcap.model.mapper import TransformerMapper, TransformerMapperWindowed
    from clipcap.model.model import ClipCapModel, ClipCapModelPrefixOnly
    from clipcap.model.config import Config, TrainingConfig
    from clipcap.encoders.config import EncoderConfig
    from clipcap.inference import base as ibase
    os.makedirs(OUT, exist_ok=True)

    # ---------------- (1) mapper: tiny + an hd=96 / S=20 case (the config-2 attention shape at small width) ----------------
    for name, (E, D, P, L, H, N, B) in {"mapper_tiny": (32, 64, 4, 4, 4, 2, 3),
                                        "mapper_hd96": (64, 192, 10, 10, 2, 2, 2)}.items():
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
