"""Edge cases of the hot path on the GPU: smallest batches / sequence lengths, ragged and fully-padded captions, odd (but legal)
widths, error codes for illegal shapes — each checked against the CPU oracle with the kernels' bf16 rounding points."""
import ctypes as C

import pytest
import torch

from oracle import clipcap_oracle as O

pytestmark = pytest.mark.gpu


def _build(E, D, P, L, H, N, n_head, n_layer, V, npos, seed=0, prec=None):
    """prec None = bf16 operands, 16 = fp16 operands (the gradients in the arena then carry the engine's loss scale), 32 = split-bf16
    operands (the reference's default precision: checked against the fp32 oracle at 50x tighter tolerances)."""
    from clipcap_amd.engine import ClipCapEngine, Gpt2Engine, MapperEngine
    torch.manual_seed(seed)
    me = MapperEngine(E, D, L, P, H, N, device="cuda", precision=prec)
    ge = Gpt2Engine(D, n_head, n_layer, V, npos, device="cuda", precision=prec)
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
    return ClipCapEngine(me, ge, train_lm=False), sd, cfg


def _check(eng, sd, cfg, tokens, embeds, tol=2e-3):
    loss = eng.forward_backward(tokens.cuda(), embeds.cuda())
    sdr = {k: v.clone().requires_grad_(k.startswith("transformer_mapper.")) for k, v in sd.items()}
    kept = int((tokens > 0).sum())
    if kept == 0:
        assert float(loss) == 0.0 and float(eng.stats[1]) == 0.0
        assert torch.count_nonzero(eng.mapper.arena.g32) == 0
        return
    fp16 = eng.scaler is not None
    x3 = eng.mapper.op_dtype == 2
    ref = O.clipcap_loss(sdr, tokens, embeds, cfg=cfg, rb=False if x3 else ("fp16" if fp16 else True))
    ref.backward()
    if x3:
        tol = min(tol, 5e-5)
    assert abs(float(loss) - float(ref)) <= tol, (float(loss), float(ref))
    unscale = 1.0 / float(eng.scaler.scale) if fp16 else 1.0
    gv = eng.mapper.views(eng.mapper.arena.g32)
    for k, v in gv.items():
        r = sdr["transformer_mapper." + k].grad
        assert ((v.cpu() * unscale - r).norm() / r.norm().clamp_min(1e-12)).item() <= (2e-3 if x3 else 6e-2), k


@pytest.mark.parametrize("prec", [None, 16, 32])
@pytest.mark.parametrize("B,cap,P,L", [(1, 1, 1, 1), (1, 5, 2, 3), (3, 2, 4, 1), (2, 9, 1, 6)])
def test_smallest_shapes(B, cap, P, L, prec):
    eng, sd, cfg = _build(16, 64, P, L, 4, 1, 4, 1, 97, 32, prec=prec)
    torch.manual_seed(B * 10 + cap)
    _check(eng, sd, cfg, torch.randint(1, 97, (B, cap)), torch.randn(B, 16))


@pytest.mark.parametrize("prec", [None, 16, 32])
def test_ragged_pads_zero_ids_and_fully_padded_rows(prec):
    eng, sd, cfg = _build(24, 64, 2, 3, 4, 2, 4, 2, 157, 40, prec=prec)
    torch.manual_seed(4)
    tokens = torch.randint(1, 157, (5, 8))
    tokens[0, 3:] = -1
    tokens[1, :] = -1              # a caption that is all padding: contributes nothing
    tokens[2, 0] = 0               # id 0 is ignored by the loss (model.py:109)
    tokens[4, 7:] = -1
    _check(eng, sd, cfg, tokens, torch.randn(5, 24))
    eng.zero_grad()
    _check(eng, sd, cfg, torch.full((2, 4), -1), torch.randn(2, 24))      # nothing kept at all: loss 0, zero gradients


@pytest.mark.parametrize("prec", [None, 16])
def test_widths_that_are_multiples_of_8_but_not_of_64(prec):
    # D = 40*... : K of the GEMMs not a multiple of 64 -> generic register-staged kernel instead of the direct-to-LDS one
    eng, sd, cfg = _build(40, 96, 3, 2, 4, 1, 4, 1, 203, 24, prec=prec)      # hd = 24, Hm = 192, E = 40
    torch.manual_seed(9)
    _check(eng, sd, cfg, torch.randint(1, 203, (3, 6)), torch.randn(3, 40), tol=3e-3)


def test_illegal_shapes_are_reported_not_miscomputed():
    from clipcap_amd import _lib
    from clipcap_amd.engine import MapperEngine
    with pytest.raises(_lib.CCError):
        MapperEngine(30, 64, 2, 2, 4, 1)           # E not a multiple of 8
    with pytest.raises(_lib.CCError):
        MapperEngine(32, 60, 2, 2, 4, 1)           # D % 8 != 0
    with pytest.raises(_lib.CCError):
        MapperEngine(32, 96, 2, 2, 8, 1)           # head dim 12: not a multiple of 8
    l = _lib.lib()
    cfg = _lib.Gpt2Cfg(64, 4, 1, 97, 128, 16)
    shp = _lib.Gpt2Shape(2, 3, 40, 37, 1)          # T > n_positions
    assert l.cc_gpt2_ws_bytes(C.byref(cfg), C.byref(shp)) == -2
    assert l.cc_mapper_fwd(None, 1, None, None, None, None, None, 0, None) == -1


def test_context_overflow_and_beam_limits():
    from types import SimpleNamespace
    from clipcap_amd.engine import DecodeSession
    from clipcap_amd.inference import generate_beam_tokens
    from clipcap_amd.model.gpt2 import GPT2LM
    lm = GPT2LM(n_embd=64, n_layer=1, n_head=4, vocab_size=97, n_positions=12).to("cuda")
    sess = DecodeSession(lm.engine, 2, 64)         # clamped to n_positions
    assert sess.ctx_max == 12
    sess.forward(torch.randn(2, 10, 64, device="cuda"))
    with pytest.raises(RuntimeError, match="overflow"):
        sess.forward(torch.randn(2, 3, 64, device="cuda"))
    toks, scores, lens = generate_beam_tokens(SimpleNamespace(language_model=lm), torch.randn(2, 4, 64, device="cuda"), beam_size=1,
                                              entry_length=5, stop_token=96)
    assert toks.shape[:2] == (2, 1) and toks.shape[2] <= 5 and torch.isfinite(scores).all()


def test_decode_row_count_paths_agree():
    """cc_decode_fwd picks its GEMM kernels by row count (<= 640 rows: 64-row tiles, with or without K split over the waves; above: the
    128-row kernels).  The same rows decoded as one 700-row session and as two 350-row sessions must agree to bf16 rounding."""
    from clipcap_amd.engine import DecodeSession
    from clipcap_amd.model.gpt2 import GPT2LM
    torch.manual_seed(5)
    lm = GPT2LM(n_embd=768, n_layer=2, n_head=12, vocab_size=1000, n_positions=64).to("cuda")
    x = torch.randn(700, 6, 768, device="cuda") * 0.4
    big = DecodeSession(lm.engine, 700, 16)
    lb = [big.forward(x[:, :5]).clone(), big.forward(x[:, 5:6]).clone()]
    for half in (slice(0, 350), slice(350, 700)):
        s = DecodeSession(lm.engine, 350, 16)
        ls = [s.forward(x[half, :5]).clone(), s.forward(x[half, 5:6]).clone()]
        for a, b in zip(ls, lb):
            d = (a - b[half]).float()
            scale = max(1.0, b.abs().max().item())
            assert d.abs().max().item() <= 8e-3 * scale and d.pow(2).mean().sqrt().item() <= 2e-3 * scale


@pytest.mark.parametrize("mode", [0, 3, 4, 5, 6, 7])
def test_whole_step_under_every_forced_gemm_tile(mode):
    """The NT tile chooser picks by size, so small tests never reach the 256- / 320-row kernels through the model: force each tile
    (cc_gemm_tile_mode) for every NT GEMM of a training step — residual, gelu', lm_head + cross-entropy epilogues included — and
    check loss and mapper gradients against the oracle."""
    from clipcap_amd import _lib
    eng, sd, cfg = _build(32, 128, 3, 5, 4, 2, 4, 2, 1000, 64)
    torch.manual_seed(mode)
    tokens = torch.randint(1, 1000, (6, 11))
    tokens[1, 7:] = -1
    old = _lib.lib().cc_gemm_tile_mode(mode)
    try:
        _check(eng, sd, cfg, tokens, torch.randn(6, 32))
    finally:
        _lib.lib().cc_gemm_tile_mode(old)


def test_lm_head_gradient_k_slices_equal_the_single_pass():
    """The lm_head input gradient runs as K slices on the 256 x 256 kernel + a slab pass from K = 4096 (vocabulary) columns up
    (gemm_nt_deepk); cc_gemm_tile_mode(0) takes the single-pass 128 x 128 path instead.  Same step, both paths, at the real vocabulary
    width: loss identical, mapper gradients equal up to the 16-bit rounding of the one tensor that differs in summation order."""
    from clipcap_amd import _lib
    eng, sd, cfg = _build(64, 256, 4, 4, 4, 1, 4, 1, 50257, 64, seed=3)
    torch.manual_seed(1)
    tokens = torch.randint(1, 50257, (6, 20)).cuda()
    tokens[2, 11:] = -1
    embeds = torch.randn(6, 64).cuda()
    eng.zero_grad()
    l_deep = float(eng.forward_backward(tokens, embeds))
    g_deep = eng.mapper.arena.g32.clone()
    old = _lib.lib().cc_gemm_tile_mode(0)
    try:
        eng.zero_grad()
        l_flat = float(eng.forward_backward(tokens, embeds))
        g_flat = eng.mapper.arena.g32.clone()
    finally:
        _lib.lib().cc_gemm_tile_mode(old)
    assert abs(l_deep - l_flat) <= 1e-6 * abs(l_flat)
    assert torch.isfinite(g_deep).all() and g_flat.abs().max() > 0
    assert ((g_deep - g_flat).norm() / g_flat.norm()).item() <= 5e-3
