"""GPT-2 train-mode dropout (full finetune): the kernels' counter-based masks are read back through cc_dropout_mask and handed to
the oracle (hf modeling_gpt2.py: embd / attention-probability / residual dropout), so loss and gradients are compared like for like."""
import ctypes as C

import pytest
import torch

from oracle import clipcap_oracle as O

pytestmark = pytest.mark.gpu


def _build(E, D, P, L, H, N, n_head, n_layer, V, npos, seed=0):
    from clipcap_amd.engine import ClipCapEngine, Gpt2Engine, MapperEngine
    torch.manual_seed(seed)
    me = MapperEngine(E, D, L, P, H, N, device="cuda")
    ge = Gpt2Engine(D, n_head, n_layer, V, npos, device="cuda")
    sd = {}
    for pre, eng in (("transformer_mapper.", me), ("language_model.", ge)):
        for k, v in eng.views(eng.arena.w32).items():
            if ("norm" in k or "ln_" in k) and k.endswith("weight"):
                t = 1.0 + 0.05 * torch.randn(v.shape)
            elif k.endswith(".bias"):
                t = 0.02 * torch.randn(v.shape)
            elif "prefix_const" in k:
                t = torch.randn(v.shape)
            else:
                t = torch.randn(v.shape) * (0.1 if ("wte" in k or "wpe" in k) else 0.5 / v.shape[-1] ** 0.5)
            sd[pre + k] = t
            v.copy_(t)
    cfg = dict(projection_length=P, prefix_length=L, heads=H, layers=N, n_head=n_head, n_layer=n_layer)
    return ClipCapEngine(me, ge, train_lm=True), sd, cfg


def _mask(seed, site, layer, p, shape):
    from clipcap_amd import _lib
    n = 1
    for s in shape:
        n *= s
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    rc = _lib.lib().cc_dropout_mask(seed, site, layer, p, n, C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    return out.cpu().float().reshape(shape)


def test_full_finetune_with_dropout_matches_oracle_with_the_same_masks():
    E, D, P, L, H, N, n_head, n_layer, V, npos = 16, 128, 2, 3, 2, 1, 2, 2, 157, 32        # GPT-2 head dim 64 (MFMA attention)
    eng, sd, cfg = _build(E, D, P, L, H, N, n_head, n_layer, V, npos)
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
    for k, p in (("embd", p_e), ("resid_mlp", p_r)):
        m = drop[k] if k == "embd" else torch.stack(drop[k])
        assert abs(float(m.mean()) - (1.0 - p)) < 0.03, (k, float(m.mean()))                # keep rate
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.clipcap_loss(sdr, tokens, embeds, cfg=cfg, rb=True, drop=drop)
    ref.backward()
    sdp = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    plain = O.clipcap_loss(sdp, tokens, embeds, cfg=cfg, rb=True)
    plain.backward()
    assert abs(float(loss) - float(ref.detach())) <= 2e-3, (float(loss), float(ref.detach()))
    discriminating = 0
    for pre, e in (("transformer_mapper.", eng.mapper), ("language_model.", eng.gpt2)):
        for k, v in e.views(e.arena.g32).items():
            r = sdr[pre + k].grad
            if "lm_head" in k or r is None:
                continue
            assert ((v.cpu() - r).norm() / r.norm().clamp_min(1e-12)).item() <= 6e-2, (pre + k)
            # the masks matter: the no-dropout gradient of the same tensor is far outside that tolerance
            discriminating += ((sdp[pre + k].grad - r).norm() / r.norm().clamp_min(1e-12)).item() > 0.2
    assert discriminating >= 20, discriminating
    # p = 0 is exactly the eval-mode step; the process-global setting does not leak out of the call
    eng.zero_grad()
    l0 = float(eng.forward_backward(tokens.cuda(), embeds.cuda(), dropout=(0.0, 0.0, 0.0, 5)))
    eng.zero_grad()
    l1 = float(eng.forward_backward(tokens.cuda(), embeds.cuda()))
    assert l0 == l1 and abs(l1 - float(plain.detach())) <= 2e-3
    eng.zero_grad()
    assert float(eng.forward_backward(tokens.cuda(), embeds.cuda(), dropout=(p_e, p_a, p_r, seed + 1))) != float(loss)


def test_dropout_in_the_sliced_backward_equals_the_single_call_backward():
    eng, sd, cfg = _build(16, 128, 2, 3, 2, 1, 2, 3, 157, 32)
    torch.manual_seed(2)
    tokens, embeds = torch.randint(1, 157, (2, 6)).cuda(), torch.randn(2, 16).cuda()
    grads = []
    for sliced in (False, True):
        eng.zero_grad()
        eng.forward_backward(tokens, embeds, dropout=(0.1, 0.1, 0.1, 77), on_grads_ready=(lambda a, lo, hi: None) if sliced else None)
        grads.append((eng.mapper.arena.g32.clone(), eng.gpt2.arena.g32.clone()))
    assert torch.allclose(grads[0][0], grads[1][0], rtol=1e-4, atol=1e-7) and torch.allclose(grads[0][1], grads[1][1], rtol=1e-4, atol=1e-7)


def test_clipcap_model_train_mode_applies_dropout_only_to_a_full_finetune():
    """ClipCapModel.train() leaves the language model in train mode -> dropout (GPT2Config pdrop defaults 0.1) is active and seeded
    from torch's generator; ClipCapModelPrefixOnly pins the language model to eval (reference model.py:120-123) -> none."""
    from clipcap_amd.encoders import EncoderConfig
    from clipcap_amd.model import ClipCapModel, ClipCapModelPrefixOnly, Config, TrainingConfig
    from clipcap_amd.model.gpt2 import GPT2LM
    tokens, embeds = torch.randint(1, 157, (2, 6)).cuda(), torch.randn(2, 16).cuda()

    def make(cls):
        torch.manual_seed(0)
        lm = GPT2LM(n_embd=128, n_layer=2, n_head=2, vocab_size=157, n_positions=32)
        cfg = Config(language_model="unused", train_language_model=(cls is ClipCapModel), prefix_length=3, projection_length=2, transformer_layers=1,
                     transformer_attention_heads=2, encoder_config=EncoderConfig(encoder_embedding_size=16),
                     training_config=TrainingConfig(optimizer_lr=0.0, use_deepspeed_optimisers=False, scheduler_warmup_steps=1, total_steps=4))
        return cls(cfg, language_model=lm).to("cuda")

    m = make(ClipCapModel).train()
    assert m._dropout() is not None
    torch.manual_seed(1)
    a = float(m.fused_step((tokens, embeds), lr=0.0))
    torch.manual_seed(1)
    b = float(m.fused_step((tokens, embeds), lr=0.0))
    c = float(m.fused_step((tokens, embeds), lr=0.0))
    assert a == b and a != c                     # same generator state -> same masks; next draw -> other masks
    m.eval()
    assert m._dropout() is None
    e1, e2 = float(m.fused_step((tokens, embeds), lr=0.0)), float(m.fused_step((tokens, embeds), lr=0.0))
    assert e1 == e2 and e1 != a
    po = make(ClipCapModelPrefixOnly).train()
    assert po._dropout() is None and not po.language_model.training
