"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol the header declares, the Python
modules expose the reference's names / state-dict keys / config schema / CLI flags, the data path honours the reference's
batch contract, and the product path refuses to run without its HIP kernels."""
import argparse
import os
import re

import numpy as np
import pytest
import torch
import yaml

from tests.util import load_golden, sd_of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeTokenizer:
    eos_token = "<eos>"
    bos_token = "<bos>"

    def __init__(self, vocab=157, eos=5):
        self.vocab, self.eos = vocab, eos

    def encode(self, s, return_tensors=None):
        if s == self.eos_token:
            return [self.eos]
        ids = [(ord(c) * 7 + i) % (self.vocab - 1) + 1 for i, c in enumerate(s)]
        return torch.tensor([ids]) if return_tensors == "pt" else ids

    def batch_encode_plus(self, caps):
        return {"input_ids": [self.encode(c) for c in caps]}

    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)


def test_library_exports_every_declared_symbol():
    from clipcap_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "clipcap_hip.h")).read()
    declared = set(re.findall(r"\b(cc_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    l = _lib.lib()
    for name in declared:
        assert hasattr(l, name), f"{name} declared in clipcap_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert l.cc_abi_version() == 3


def _nm(path, *flags):
    import subprocess
    out = subprocess.run(["nm", "-D", *flags, path], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def test_library_exports_exactly_the_header_and_nothing_else():
    """the product library exports the entry points of include/clipcap_hip.h and nothing else (csrc/exports.map, generated
    by tools/gen_abi.py) — the per-operand-type variants (*_bf16 / *_f16 / *_x3) and the C++ internals are local; it never reads the
    environment (clipcap_amd/csrc/lab_env.h: getenv is not even imported) and carries none of the experiment kernels, which live in the
    lab build (`make lab` -> libclipcap_hip_lab.so; tests/test_gpu_lab.py)."""
    import shutil
    if shutil.which("nm") is None:
        pytest.skip("binutils nm not available")
    hdr = open(os.path.join(ROOT, "include", "clipcap_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"^(?:int|int64_t)\s+(cc_\w+)\s*\(", hdr, flags=re.M))
    lib = os.path.join(ROOT, "clipcap_amd", "libclipcap_hip.so")
    exported = {n for n in _nm(lib, "--defined-only") if not n.startswith("_") or n.startswith("cc_")}
    exported = {n for n in exported if n not in ("_init", "_fini", "_edata", "_end", "__bss_start")}
    assert exported == declared, (sorted(exported - declared)[:10], sorted(declared - exported)[:10])
    assert "getenv" not in _nm(lib, "--undefined-only"), "the product library must not read the environment"
    blob = open(lib, "rb").read()
    for marker in (b"k_decode_xt", b"k_decode_layers", b"k_attn_fwd_f32mfma", b"gemm_nt_s64kwb_kernel", b"k_skinny_image"):
        assert marker not in blob, f"experiment kernel {marker!r} in the product library"
    # (round 6) the default-off decode experiments' entry points are declared in include/clipcap_hip_lab.h and exported by the lab library only
    lhdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "clipcap_hip_lab.h")).read(), flags=re.S)
    lab_declared = set(re.findall(r"^(?:int|int64_t)\s+(cc_\w+)\s*\(", lhdr, flags=re.M))
    assert lab_declared and not (lab_declared & declared)
    for name in ("cc_decode_fwd_x", "cc_decode_image", "cc_decode_xt_image", "cc_decode_ws_check", "cc_decode_last_path"):
        assert name in lab_declared and name not in exported
    lab = os.path.join(ROOT, "clipcap_amd", "libclipcap_hip_lab.so")
    if os.path.exists(lab):
        assert {n for n in _nm(lab, "--defined-only") if n.startswith("cc_")} == declared | lab_declared
        lblob = open(lab, "rb").read()
        assert b"k_decode_xt" in lblob and b"k_decode_layers" in lblob and b"k_attn_fwd_f32mfma" in lblob and b"gemm_nt_s64kwb_kernel" in lblob


def test_abi_dispatch_file_is_current():
    """clipcap_amd/csrc/abi_dispatch.cpp is generated from include/clipcap_hip.h (tools/gen_abi.py): stale output = missing symbols."""
    import subprocess
    import sys
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_abi.py"), "--check"]).returncode == 0


def test_param_layout_matches_reference_counts():
    from clipcap_amd.engine import Gpt2Engine, MapperEngine
    me = MapperEngine(512, 768, 10, 10, 8, 8)
    assert me.arena.n == 41_745_408            # reference TransformerMapper parameter count
    ge = Gpt2Engine(768, 12, 12, 50257, 1024)
    assert ge.arena.n == 124_439_808 + (50304 - 50257) * 768   # HF gpt2 + zero vocab padding rows
    # views tile the arena without overlap
    spans = sorted((off, off + int(np.prod(shape))) for _, off, shape in me.shapes())
    assert spans[0][0] == 0 and spans[-1][1] == me.arena.n
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_state_dict_keys_are_the_references():
    from clipcap_amd.encoders import EncoderConfig
    from clipcap_amd.model import ClipCapModel, ClipCapModelPrefixOnly, Config
    from clipcap_amd.model.gpt2 import GPT2LM
    g = load_golden("train_prefix_only")
    ref_keys = set(sd_of(g))
    lm = GPT2LM(n_embd=64, n_layer=2, n_head=4, vocab_size=157, n_positions=40)
    cfg = Config(language_model="unused", prefix_length=3, projection_length=2, transformer_layers=2, transformer_attention_heads=4,
                 encoder_config=EncoderConfig(encoder_embedding_size=24))
    m = ClipCapModelPrefixOnly(cfg, language_model=lm)
    assert set(m.state_dict()) == ref_keys
    res = m.load_state_dict(sd_of(g), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    # PrefixOnly.parameters() is the mapper only (model.py:117-118); the full model exposes everything
    assert sum(p.numel() for p in m.parameters()) == sum(v.size for k, v in g.items() if k.startswith("sd.transformer_mapper."))
    assert sum(p.numel() for p in ClipCapModel.parameters(m)) > sum(p.numel() for p in m.parameters())
    # .train() keeps the LM in eval (model.py:120-123)
    m.train()
    assert m.training and not m.language_model.training
    # tied head
    assert m.language_model.lm_head.weight is m.language_model.get_input_embeddings().weight
    assert m.lm_embedding_size == 64


def test_windowed_mapper_keys_and_shapes():
    from clipcap_amd.model.mapper import TransformerMapperWindowed
    g = load_golden("mapper_windowed")
    E, D, P, L, H, N, B, W = [int(v) for v in g["dims"]]
    m = TransformerMapperWindowed(E, D, L, P, W, True, H, N)
    assert set(m.state_dict()) == set(sd_of(g))
    m.load_state_dict(sd_of(g), strict=True)


def test_config_yaml_roundtrip_and_reference_schema(tmp_path):
    from clipcap_amd.encoders import EncoderConfig
    from clipcap_amd.model import Config, TrainingConfig
    cfg = Config(language_model="gpt2", encoder_config=EncoderConfig(encoder_embedding_size=512), training_config=TrainingConfig())
    d = cfg.to_dict()
    assert list(d) == ["language_model", "train_language_model", "prefix_length", "projection_length", "transformer_layers",
                       "transformer_attention_heads", "use_positional_embeddings", "encoder_config", "training_config"]
    assert list(d["encoder_config"]) == ["encoder_model_name", "encoder_model_variant", "encoder_embedding_size", "normalize_embeddings",
                                         "use_windowed_embeddings", "window_size", "window_overlap_percentage"]
    assert d["transformer_attention_heads"] == 16 and d["encoder_config"]["window_size"] == 16
    p = tmp_path / "c.yaml"
    p.write_text(yaml.dump(d))
    raw = yaml.safe_load(p.read_text())
    raw["encoder_config"] = EncoderConfig(**raw["encoder_config"])
    raw["training_config"] = TrainingConfig(**raw["training_config"])
    assert Config(**raw) == cfg


def test_cli_flags_match_reference_defaults():
    from clipcap_amd.model import add_model_args
    from clipcap_amd.train import add_training_args
    ns = add_model_args(add_training_args(argparse.ArgumentParser())).parse_args([])
    expect = dict(batch_size=64, epochs=5, optimizer_lr=2e-5, scheduler_warmup_steps=5000, fp_precision=32, checkpoint_save_frequency=1,
                  checkpoint_filename_prefix=1, device="0", input_dataset="./dataset/", output_folder="./models/", reader_max_piece_size=50,
                  reader_parallel_pieces=10, enable_deepspeed=False, deepspeed_strategy=None, enable_wandb=False, wandb_project="clipcap",
                  logging_frequency=50, language_model="gpt2-xl", prefix_length=10, projection_length=10, train_language_model=False,
                  transformer_layers=8, transformer_attention_heads=8, use_positional_embeddings=True)
    got = vars(ns)
    extensions = {"resume_from": None}                     # flags this trainer adds (the reference cannot resume, train.py:17-93)
    assert {k: v for k, v in got.items() if k not in extensions} == expect
    assert {k: got[k] for k in extensions} == extensions


def _write_dataset(path, n=23, E=24, shards=(10, 13)):
    import pyarrow as pa
    import pyarrow.parquet as pq
    os.makedirs(path / "embeddings")
    os.makedirs(path / "captions")
    rng = np.random.default_rng(0)
    emb = rng.standard_normal((n, E)).astype(np.float32)
    caps = [f"caption number {i} " + "x" * (i % 7) for i in range(n)]
    lo = 0
    for i, c in enumerate(shards):
        np.save(path / "embeddings" / f"embeds_{i:05d}.npy", emb[lo:lo + c])
        pq.write_table(pa.table({"caption": caps[lo:lo + c]}), path / "captions" / f"captions_{i:05d}.parquet")
        lo += c
    (path / "encoder_config.yaml").write_text(yaml.dump(dict(encoder_model_name="clip", encoder_model_variant="ViT-L/14",
                                                             encoder_embedding_size=None, normalize_embeddings=False,
                                                             use_windowed_embeddings=False, window_size=16, window_overlap_percentage=0.0)))
    return emb, caps


def test_dataloader_batch_contract_and_rank_sharding(tmp_path):
    from clipcap_amd.train.dataloader import EmbedDataset
    emb, caps = _write_dataset(tmp_path)
    tok = FakeTokenizer()
    ds = EmbedDataset(str(tmp_path), batch_size=8, max_token_length=20, tokenizer=tok)
    assert ds.encoder_embedding_size == 24 and len(ds) == 3
    batches = list(ds)
    assert [b[0].shape[0] for b in batches] == [8, 8, 7]          # crosses the shard boundary at row 10
    tokens = torch.cat([b[0] for b in batches])
    embeds = torch.cat([b[1] for b in batches])
    assert tokens.dtype == torch.int64 and tokens.shape == (23, 20) and embeds.dtype == torch.float32
    assert np.array_equal(embeds.numpy(), emb)
    for i, c in enumerate(caps):                                   # -1 right padding / truncation (dataloader.py:41-50)
        ids = tok.encode(c)[:20]
        assert tokens[i, :len(ids)].tolist() == ids and (tokens[i, len(ids):] == -1).all()
    # two ranks see disjoint halves of every global batch, together the whole dataset
    r = [list(EmbedDataset(str(tmp_path), batch_size=4, max_token_length=20, tokenizer=tok, rank=k, world_size=2)) for k in (0, 1)]
    seen = torch.cat([torch.cat([b[1] for b in r[0]]), torch.cat([b[1] for b in r[1]])])
    assert seen.shape[0] == 23 and len({tuple(np.round(row.numpy(), 5)) for row in seen}) == 23
    assert len(r[0]) == len(r[1]) == EmbedDataset(str(tmp_path), batch_size=4, tokenizer=tok, rank=0, world_size=2).__len__()
    # a trailing global batch with fewer rows than ranks (23 % (11*2) == 1) is dropped on EVERY rank, so no rank skips a
    # step whose collectives the others enter; len() agrees with the number of batches on both ranks
    t = [EmbedDataset(str(tmp_path), batch_size=11, max_token_length=20, tokenizer=tok, rank=k, world_size=2) for k in (0, 1)]
    assert [len(list(d)) for d in t] == [1, 1] and len(t[0]) == len(t[1]) == 1
    # ... while a tail with at least one row per rank is kept everywhere (23 % (7*3) == 2 < 3 dropped; 23 % (5*2) == 3 kept)
    t = [EmbedDataset(str(tmp_path), batch_size=5, max_token_length=20, tokenizer=tok, rank=k, world_size=2) for k in (0, 1)]
    assert [len(list(d)) for d in t] == [3, 3] and len(t[0]) == 3 and sum(b[0].shape[0] for d in t for b in d) == 23


def test_schedule_matches_oracle():
    from clipcap_amd.model.optim import linear_warmup_decay
    from oracle.clipcap_oracle import linear_schedule_factor
    f = linear_warmup_decay(7, 40)
    assert all(abs(f(s) - linear_schedule_factor(s, 7, 40)) < 1e-15 for s in range(60))
    g = load_golden("train_prefix_only")
    f = linear_warmup_decay(2, 6)
    assert np.allclose([1e-3 * f(s) for s in range(3)], g["lrs"])


def test_no_cpu_fallback_and_loud_failure(monkeypatch):
    from clipcap_amd import _lib
    from clipcap_amd.engine import MapperEngine
    eng = MapperEngine(32, 64, 4, 4, 4, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        eng.forward(torch.randn(2, 32))
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libclipcap_hip.so")
    with pytest.raises(_lib.HipExtensionMissing):
        _lib.lib()


def test_product_never_imports_the_oracle():
    import subprocess
    import sys
    code = ("import sys; import clipcap_amd, clipcap_amd.model, clipcap_amd.train, clipcap_amd.inference, clipcap_amd.engine; "
            "bad=[m for m in sys.modules if m.startswith('oracle')]; assert not bad, bad")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "clipcap_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_checkpoint_files_follow_reference_naming_and_load(tmp_path):
    from clipcap_amd.encoders import EncoderConfig
    from clipcap_amd.model import ClipCapModelPrefixOnly, Config, TrainingConfig, load
    from clipcap_amd.model.gpt2 import GPT2LM
    from clipcap_amd.train.callback import CheckpointSaver
    lm = GPT2LM(n_embd=64, n_layer=2, n_head=4, vocab_size=157, n_positions=40)
    lm_dir = tmp_path / "lm"
    lm.save_pretrained(str(lm_dir))
    cfg = Config(language_model=str(lm_dir), prefix_length=3, projection_length=2, transformer_layers=2, transformer_attention_heads=4,
                 encoder_config=EncoderConfig(encoder_embedding_size=24), training_config=TrainingConfig(total_steps=3))
    m = ClipCapModelPrefixOnly(cfg)
    assert torch.equal(m.language_model.state_dict()["transformer.h.1.mlp.c_fc.weight"], lm.state_dict()["transformer.h.1.mlp.c_fc.weight"])
    saver = CheckpointSaver(str(tmp_path / "out"), "demo")
    saver.save_config(cfg.to_dict())
    saver.on_epoch_end(m, 0)
    saver.save_final_checkpoint(m)
    assert sorted(os.listdir(tmp_path / "out")) == ["demo_config.yaml", "demo_epoch_0.ckpt", "demo_final.ckpt"]
    m2, tok = load(str(tmp_path / "out" / "demo_final.ckpt"), str(tmp_path / "out" / "demo_config.yaml"), device="cpu", from_checkpoint=True,
                   tokenizer=FakeTokenizer())
    assert isinstance(m2, ClipCapModelPrefixOnly) and not m2.training and m2.config.training_config is None
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k


def test_sampling_filters_match_reference_outputs():
    """clipcap_amd.inference.utils vs the reference's filter outputs captured in tests/golden/filters.npz."""
    from clipcap_amd.inference import utils as U
    g = load_golden("filters")
    lg = torch.from_numpy(g["in.logits"])
    toks = torch.from_numpy(g["in.tokens"])
    assert np.array_equal(U.top_k_top_p_filtering(lg, top_k=5).numpy(), g["topk5"])
    assert np.array_equal(U.top_k_top_p_filtering(lg, top_p=0.8).numpy(), g["topp08"])
    assert np.array_equal(U.top_k_top_p_filtering(lg, top_k=10, top_p=0.5).numpy(), g["topk10_topp05"])
    assert np.allclose(U.repetition_penalty_apply(lg, toks, 1.2).numpy(), g["rep12"], atol=1e-6)
    assert np.allclose(U.sentence_length_penalty_apply(torch.from_numpy(g["in.logits2"]), toks, 11, 4, 50, 1.0).numpy(), g["lenpen"], atol=1e-6)
    # batched use (the reference is 1-D only): rows are filtered independently
    two = torch.stack((lg, lg.flip(0)))
    out = U.top_k_top_p_filtering(two, top_k=5)
    assert np.array_equal(out[0].numpy(), g["topk5"]) and (out[1] > -float("inf")).sum() == 5
    # nucleus distribution vs the captured final_p (logits recomputed by the oracle from the same tiny GPT-2)
    from oracle import clipcap_oracle as O
    b = load_golden("beam_tiny")
    D, n_layer, n_head, V, npos = [int(v) for v in b["cfg"]]
    logits = O.gpt2_logits(sd_of(b), torch.from_numpy(g["nucleus.prefix"]), n_head, n_layer)[:, -1, :]
    assert np.allclose(U.nucleus_distribution(logits, top_p=0.8).numpy(), g["nucleus.final_p"], atol=1e-6)


def test_trim_padding_is_exact_on_the_oracle():
    """Dropping all-padding tail columns leaves the reference loss and gradients unchanged (oracle check of the data-path trim)."""
    from clipcap_amd.train.dataloader import trim_padding
    from oracle import clipcap_oracle as O
    g = load_golden("train_prefix_only")
    E, D, P, L, H, N, n_head, n_layer, V, npos = [int(v) for v in g["cfg"]]
    cfg = dict(projection_length=P, prefix_length=L, heads=H, layers=N, n_head=n_head, n_layer=n_layer)
    sd = sd_of(g)
    sd.pop("language_model.lm_head.weight", None)
    torch.manual_seed(2)
    tokens = torch.full((4, 24), -1, dtype=torch.int64)
    for b, n in enumerate((5, 9, 3, 11)):
        tokens[b, :n] = torch.randint(1, V, (n,))
    embeds = torch.randn(4, E)
    trimmed = trim_padding(tokens)
    assert trimmed.shape == (4, 16) and torch.equal(trimmed, tokens[:, :16])
    names = [k for k in sd if k.startswith("transformer_mapper.")]
    outs = []
    for tk in (tokens, trimmed):
        sdr = {k: v.clone().requires_grad_(k in names) for k, v in sd.items()}
        loss = O.clipcap_loss(sdr, tk, embeds, cfg=cfg)
        loss.backward()
        outs.append((float(loss.detach()), torch.cat([sdr[k].grad.flatten() for k in names])))
    assert abs(outs[0][0] - outs[1][0]) <= 1e-6
    assert (outs[0][1] - outs[1][1]).abs().max().item() <= 1e-6


def test_enable_deepspeed_selects_fusedadam_weight_decay():
    """reference model.py:72-77: use_deepspeed_optimisers -> deepspeed FusedAdam(adam_w_mode=True) with ITS default weight decay
    (0.0); otherwise torch.optim.AdamW (0.01).  The flag has no other effect here, but that one is followed."""
    from clipcap_amd.encoders import EncoderConfig
    from clipcap_amd.model import ClipCapModelPrefixOnly, Config, TrainingConfig
    from clipcap_amd.model.gpt2 import GPT2LM
    for ds, wd in ((True, 0.0), (False, 0.01)):
        lm = GPT2LM(n_embd=64, n_layer=1, n_head=2, vocab_size=97, n_positions=16)
        cfg = Config(language_model="unused", train_language_model=False, prefix_length=2, projection_length=2, transformer_layers=1,
                     transformer_attention_heads=2, encoder_config=EncoderConfig(encoder_embedding_size=16),
                     training_config=TrainingConfig(optimizer_lr=1e-3, use_deepspeed_optimisers=ds, scheduler_warmup_steps=1, total_steps=4))
        m = ClipCapModelPrefixOnly(cfg, language_model=lm)
        assert m._weight_decay() == wd
        assert m.configure_optimizers()["optimizer"].param_groups[0]["weight_decay"] == wd


def test_precision_flag_maps_to_operand_modes():
    """--fp-precision as the reference defines it (train/args.py:30-34): 32 (default) / 64 -> split-bf16 operands, 16 -> fp16,
    plus this build's 'bf16'."""
    from clipcap_amd._lib import OP_BF16, OP_FP16, OP_X3, op_dtype_of
    from clipcap_amd.train import add_training_args
    assert [op_dtype_of(v) for v in (32, 64, 16, "bf16", None)] == [OP_X3, OP_X3, OP_FP16, OP_BF16, OP_BF16]
    ap = add_training_args(argparse.ArgumentParser())
    assert ap.parse_args([]).fp_precision == 32 and ap.parse_args(["--fp-precision", "bf16"]).fp_precision == "bf16"
    assert ap.parse_args(["--fp-precision", "16"]).fp_precision == 16
    with pytest.raises(ValueError):
        op_dtype_of(8)


def test_gradient_wire_dtype_follows_the_operand_mode():
    """the fp32-parity mode (split-bf16 operands, the CLI default) must not round its gradients to bf16 on the wire."""
    from clipcap_amd._lib import OP_BF16, OP_FP16, OP_X3
    from clipcap_amd.train.train import grad_wire_dtype
    assert grad_wire_dtype(OP_X3, False) is torch.float32 and grad_wire_dtype(OP_X3, True) is torch.float32
    assert grad_wire_dtype(OP_BF16, False) is torch.bfloat16 and grad_wire_dtype(OP_FP16, False) is torch.bfloat16
    assert grad_wire_dtype(OP_BF16, True) is torch.float32 and grad_wire_dtype(OP_FP16, True) is torch.float32
