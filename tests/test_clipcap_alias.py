"""The reference's import surface under its own name: users of the reference type ``import clipcap``,
``clipcap.load``, ``clipcap.inference.base.generate_beam``, ``python -m clipcap.train`` (/root/reference/clipcap/__init__.py:1-2,
train/__main__.py:1-4, docs/inference.md:13-34).  clipcap_amd.install_as_clipcap() — or the opt-in shim/ directory on PYTHONPATH — makes
those names resolve to this package."""
import json
import os
import subprocess
import sys
import types

import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def clean_alias():
    saved = {k: v for k, v in sys.modules.items() if k == "clipcap" or k.startswith("clipcap.")}
    for k in saved:
        del sys.modules[k]
    yield
    for k in [k for k in sys.modules if k == "clipcap" or k.startswith("clipcap.")]:
        del sys.modules[k]
    sys.modules.update(saved)


def _byte_level_tokenizer_files(d):
    """A GPT-2 tokenizer that needs no download: the 256 byte-level symbols + <|endoftext|>, no merges."""
    # GPT-2's byte <-> printable-character table: printable latin-1 bytes map to themselves, the rest to 256, 257, ...
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    chars, n = {}, 0
    for b in range(256):
        if b in keep:
            chars[b] = chr(b)
        else:
            chars[b] = chr(256 + n)
            n += 1
    vocab = {chars[b]: b for b in range(256)}
    vocab["<|endoftext|>"] = len(vocab)
    with open(os.path.join(d, "vocab.json"), "w") as f:
        json.dump(vocab, f)
    with open(os.path.join(d, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n")
    return len(vocab)


def _write_model(tmp_path, device="cpu"):
    """./model.pt + ./model_config.yaml the way the reference's training run leaves them (callback.py: state dict + config yaml)."""
    from clipcap_amd.encoders import EncoderConfig
    from clipcap_amd.model import ClipCapModelPrefixOnly, Config
    from clipcap_amd.model.gpt2 import GPT2LM
    lm_dir = tmp_path / "lm"
    os.makedirs(lm_dir)
    V = _byte_level_tokenizer_files(str(lm_dir))
    torch.manual_seed(5)
    lm = GPT2LM(n_embd=64, n_layer=2, n_head=4, vocab_size=V, n_positions=128)
    lm.save_pretrained(str(lm_dir))
    cfg = Config(language_model=str(lm_dir), prefix_length=3, projection_length=2, transformer_layers=2, transformer_attention_heads=4,
                 encoder_config=EncoderConfig(encoder_model_name="clip", encoder_model_variant="ViT-L_14", encoder_embedding_size=24))
    m = ClipCapModelPrefixOnly(cfg)
    torch.save(m.state_dict(), str(tmp_path / "model.pt"))
    with open(tmp_path / "model_config.yaml", "w") as f:
        yaml.dump(cfg.to_dict(), f)
    return m


def test_install_as_clipcap_registers_the_reference_names(clean_alias):
    import clipcap_amd
    clipcap_amd.install_as_clipcap()
    clipcap_amd.install_as_clipcap()                       # idempotent
    import clipcap
    import clipcap.inference.base
    import clipcap.inference.generate
    import clipcap.inference.no_beam
    import clipcap.inference.nucleus_sampling
    import clipcap.encoders.config
    from clipcap.model import ClipCapModel, ClipCapModelPrefixOnly, Config, TrainingConfig, add_model_args, load  # noqa: F401  (clipcap/model/__init__.py:1-4)
    from clipcap.train import add_training_args, start_training, train  # noqa: F401  (clipcap/train/__init__.py:1-2)
    from clipcap.encoders import EncoderConfig, get_encoder, get_encoder_from_config, get_encoder_from_model  # noqa: F401
    assert clipcap is clipcap_amd and clipcap.model is clipcap_amd.model
    assert clipcap.inference.base.generate_beam is clipcap_amd.inference.base.generate_beam
    assert callable(clipcap.load) and callable(clipcap.get_encoder) and callable(clipcap.get_encoder_from_model)
    import importlib.util
    assert importlib.util.find_spec("clipcap.train.__main__") is not None        # what `python -m clipcap.train` resolves


def test_install_refuses_to_shadow_an_imported_clipcap(clean_alias):
    import clipcap_amd
    other = types.ModuleType("clipcap")
    other.__file__ = "/somewhere/else/clipcap/__init__.py"
    sys.modules["clipcap"] = other
    sys.modules["clipcap.model"] = types.ModuleType("clipcap.model")
    with pytest.raises(RuntimeError, match="already imported"):
        clipcap_amd.install_as_clipcap()
    assert sys.modules["clipcap"] is other
    clipcap_amd.install_as_clipcap(force=True)
    assert sys.modules["clipcap"] is clipcap_amd and sys.modules["clipcap.model"] is clipcap_amd.model


def test_shim_directory_makes_python_dash_m_clipcap_train_work():
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "shim"), ROOT]))
    out = subprocess.run([sys.executable, "-m", "clipcap.train", "--help"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    for flag in ("--batch-size", "--fp-precision", "--input-dataset", "--prefix-length", "--enable-deepspeed"):      # train/args.py, model/args.py
        assert flag in out.stdout, flag
    out = subprocess.run([sys.executable, "-c", "import clipcap, clipcap.inference.base as b; print(clipcap.__name__, b.generate_beam.__module__)"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.split() == ["clipcap_amd", "clipcap_amd.inference.base"], (out.stdout, out.stderr[-2000:])


def test_documented_load_and_encoder_calls_on_cpu(clean_alias, tmp_path, monkeypatch):
    """docs/inference.md:13-22 up to the point that needs the GPU: load by the two file names, the encoder pair through the hook."""
    import clipcap_amd
    from clipcap_amd.encoders import base as enc
    clipcap_amd.install_as_clipcap()
    import clipcap
    m = _write_model(tmp_path)
    monkeypatch.chdir(tmp_path)
    model, tokenizer = clipcap.load("./model.pt", "./model_config.yaml", device="cpu")
    assert type(model).__name__ == "ClipCapModelPrefixOnly" and not model.training
    for k, v in m.state_dict().items():
        assert torch.equal(v, model.state_dict()[k]), k
    assert tokenizer.encode(tokenizer.eos_token) == [256] and tokenizer.decode([104, 105]) == "hi"
    with pytest.raises(ValueError, match="invalid encoder name"):                 # nothing registered: the reference's error type (base.py:26)
        clipcap.get_encoder_from_model(model, device="cpu")
    seen = {}

    def factory(variant, **kw):
        seen.update(kw, variant=variant)
        return (lambda x: x.mean(dim=(2, 3))), (lambda path: torch.ones(3, 4, 4))

    monkeypatch.setitem(enc._FACTORIES, "clip", factory)
    encode_fn, preprocess = clipcap.get_encoder_from_model(model, device="cpu")
    assert seen["variant"] == "ViT-L/14" and seen["device"] == "cpu" and seen["normalize_embeddings"] is False     # base.py:30-31: '_' -> '/'
    assert encode_fn(preprocess("./image.jpg").unsqueeze(0)).shape == (1, 3)


@pytest.mark.gpu
def test_documented_inference_flow_under_the_reference_names(clean_alias, tmp_path, monkeypatch):
    """The call sequence of docs/inference.md:13-34 with only the module name the reference documents: load -> encoder pair ->
    preprocess / encode -> model.transformer_mapper -> clipcap.inference.base.generate_beam, on cuda:0."""
    import clipcap_amd
    from clipcap_amd.encoders import base as enc
    clipcap_amd.install_as_clipcap()
    import clipcap
    _write_model(tmp_path)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setitem(enc._FACTORIES, "clip", lambda variant, **kw: ((lambda x: x.flatten(1)[:, :24].contiguous()),
                                                                       (lambda path: torch.linspace(-1, 1, 48).view(3, 4, 4))))
    device = "cuda:0"
    model, tokenizer = clipcap.load("./model.pt", "./model_config.yaml", device=device)
    encode_fn, preprocess = clipcap.get_encoder_from_model(model, device=device)
    sample = preprocess("./image.jpg").unsqueeze(0).to(device)
    with torch.no_grad():
        embedding = encode_fn(sample)
        embedding_prefix = model.transformer_mapper(embedding)
    captions = clipcap.inference.base.generate_beam(model, tokenizer, embedding_prefix)
    assert isinstance(captions, list) and len(captions) == 1 and isinstance(captions[0], str)
    again = clipcap.inference.base.generate_beam(model, tokenizer, embedding_prefix, entry_length=67, beam_size=5)
    assert again == captions                                                      # the documented call uses the defaults (base.py:55-64)
