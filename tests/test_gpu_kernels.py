"""GPU parity of the individual HIP kernels, called through the C ABI (ctypes), against torch fp32 math on the SAME
16-bit-rounded inputs — every test runs for both operand types of the library (bf16, and fp16 = the reference's --fp-precision 16).
Asymmetric random operands everywhere (a transposed result cannot pass)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from clipcap_amd import _lib
    return _lib.lib()


def _p(t):
    return C.c_void_p(t.data_ptr())


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


OP = [0, torch.bfloat16]       # [cfg.op_dtype value, torch dtype] of the running parametrisation


@pytest.fixture(autouse=True, params=["bf16", "fp16"])
def op_dtype(request):
    OP[0], OP[1] = (0, torch.bfloat16) if request.param == "bf16" else (1, torch.float16)
    yield request.param
    OP[0], OP[1] = 0, torch.bfloat16


def _bf(t):
    return t.to(OP[1])


@pytest.mark.parametrize("al,bl", [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 192), (264, 72, 520), (8, 8, 8), (1000, 384, 1024)])
def test_gemm_layouts(al, bl, M, N, K):
    torch.manual_seed(M * 7 + N * 3 + K + al * 2 + bl)
    dev = "cuda"
    A = _bf(torch.randn(M, K, device=dev))
    B = _bf(torch.randn(K, N, device=dev) * 0.5 + 0.1)
    bias = torch.randn(N, device=dev)
    ref = A.float() @ B.float() + bias
    Ast = A.t().contiguous() if al else A.contiguous()            # al=1: stored [K][M]
    Bst = B.contiguous() if bl else B.t().contiguous()            # bl=0: stored [N][K]
    Cm = torch.full((M, N + 8), float("nan"), device=dev)
    rc = _lib().cc_gemm_op16_f32(OP[0], al, bl, _p(Ast), Ast.shape[1], _p(Bst), Bst.shape[1], M, N, K, _p(Cm), N + 8, _p(bias), 1, _st())
    assert rc == 0
    torch.cuda.synchronize()
    err = (Cm[:, :N] - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item() / 10), err
    assert torch.isnan(Cm[:, N:]).all()   # nothing written outside the N columns


@pytest.mark.parametrize("mode", [3, 4, 5, 6, 7])
@pytest.mark.parametrize("M,N,K", [(256, 256, 32), (8, 8, 32), (520, 200, 96), (1000, 392, 1024), (300, 776, 160), (640, 512, 64), (330, 248, 128),
                                   (520, 776, 768), (161, 264, 384), (300, 520, 256), (6400, 2048, 384), (3000, 768, 2304)])
def test_gemm_nt_256_row_tiles(mode, M, N, K):
    """the 256 x 192 / 256 x 256 / 320 x 256 / 160 x 256 (mode 6) group-staggered NT kernels (forced: the chooser would pick 128 x 128 at these sizes)
    and (mode 7) the persistent 4-wave 160 x 256 kernel with the instruction-level K loop (gemm_q4.hip.h): three 64-deep stages for K % 192 == 0,
    two for K % 128 == 0 (shortest legal K of either ring included), the staggered kernel for any other K; 6400 x 2048 = 320 tiles > one per CU, so
    workgroups walk more than one tile and the cross-tile prefetch is live."""
    torch.manual_seed(M + N + K + mode)
    dev = "cuda"
    A = _bf(torch.randn(M, K, device=dev))
    Bt = _bf(torch.randn(N, K, device=dev) * 0.5 + 0.1)
    bias = torch.randn(N, device=dev)
    ref = A.float() @ Bt.float().t() + bias
    Cm = torch.full((M, N + 8), float("nan"), device=dev)
    old = _lib().cc_gemm_tile_mode(mode)
    try:
        rc = _lib().cc_gemm_op16_f32(OP[0], 0, 0, _p(A), K, _p(Bt), K, M, N, K, _p(Cm), N + 8, _p(bias), 1, _st())
        torch.cuda.synchronize()
    finally:
        _lib().cc_gemm_tile_mode(old)
    assert rc == 0
    err = (Cm[:, :N] - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item() / 10), err
    assert torch.isnan(Cm[:, N:]).all()


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
@pytest.mark.parametrize("M,N,K,ks", [(320, 1024, 1024, 1), (320, 3072, 1024, 1), (5, 2304, 768, 3), (333, 1000, 256, 1), (64, 64, 64, 1),
                                      (640, 4096, 1024, 1), (77, 200, 4096, 6), (1, 8, 128, 2), (320, 4096, 1024, 1), (81, 72, 192, 1)])
def test_gemm_nt_skinny_64_row_tiles(mode, M, N, K, ks):
    """the 64 x 64 / 64 x 128 decode-sized NT kernels (32x32x16 MFMA; mode 3: the 64 x 64 form with K split over the waves; mode 4: its
    80 x 64 form, four row tiles for M = 320), single pass and with K slices (atomic accumulation into a non-zero C), ragged M / N edges and
    a padded leading dimension"""
    torch.manual_seed(M + N + K + mode)
    dev = "cuda"
    A = _bf(torch.randn(M, K, device=dev))
    Bt = _bf(torch.randn(N, K, device=dev) * 0.5 + 0.1)
    bias = torch.randn(N, device=dev)
    C0 = torch.randn(M, N + 8, device=dev)
    ref = A.float() @ Bt.float().t() + (bias if ks == 1 else C0[:, :N])
    Cm = C0.clone()
    old = _lib().cc_gemm_skinny_mode(mode)
    try:
        rc = _lib().cc_gemm_op16_f32(OP[0], 0, 0, _p(A), K, _p(Bt), K, M, N, K, _p(Cm), N + 8, _p(bias), ks, _st())
        torch.cuda.synchronize()
    finally:
        _lib().cc_gemm_skinny_mode(old)
    assert rc == 0
    err = (Cm[:, :N] - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item() / 10), err
    assert torch.equal(Cm[:, N:], C0[:, N:])


@pytest.mark.parametrize("mode", [4, 0, -1])
@pytest.mark.parametrize("K,Mw,Nw", [(5120, 768, 1536), (1024, 264, 200), (96, 8, 8), (2080, 520, 776), (12800, 768, 768), (1237, 192, 264)])
def test_gemm_wgrad_kernels(mode, K, Mw, Nw):
    """dW += X^T Y through cc_gemm_wgrad: the 256 x 256 DMA + transpose-read kernel (mode 4 forces it wherever K % 32 == 0; mode -1 picks it from
    ~30 GFLOP) and the register-staged 128 x 128 kernel (mode 0, and any K); accumulation into a non-zero dW with a padded leading dimension."""
    torch.manual_seed(K + Mw + Nw)
    dev = "cuda"
    X = _bf(torch.randn(K, Mw, device=dev))
    Y = _bf(torch.randn(K, Nw, device=dev) * 0.5)
    dW0 = torch.randn(Mw, Nw + 4, device=dev)
    dW = dW0.clone()
    ref = dW0[:, :Nw] + X.float().t() @ Y.float()
    scratch = torch.empty(_lib().cc_wgrad_scratch_bytes(), dtype=torch.uint8, device=dev)
    old = _lib().cc_gemm_tile_mode(mode)
    try:
        rc = _lib().cc_gemm_wgrad(OP[0], _p(X), Mw, _p(Y), Nw, Mw, Nw, K, _p(dW), Nw + 4, _p(scratch), _st())
        torch.cuda.synchronize()
    finally:
        _lib().cc_gemm_tile_mode(old)
    assert rc == 0
    err = (dW[:, :Nw] - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item() / 10), err
    assert torch.equal(dW[:, Nw:], dW0[:, Nw:])


@pytest.mark.parametrize("ksplit", [2, 5, 16])
def test_gemm_wgrad_split_k_atomic(ksplit):
    torch.manual_seed(ksplit)
    dev = "cuda"
    Kk, Mw, Nw = 1237, 192, 264      # K (rows of the activations) is free: not a multiple of anything
    X = _bf(torch.randn(Kk, Mw, device=dev))
    Y = _bf(torch.randn(Kk, Nw, device=dev))
    ref = X.float().t() @ Y.float()
    Cm = torch.zeros(Mw, Nw, device=dev)
    rc = _lib().cc_gemm_op16_f32(OP[0], 1, 1, _p(X), Mw, _p(Y), Nw, Mw, Nw, Kk, _p(Cm), Nw, None, ksplit, _st())
    assert rc == 0
    torch.cuda.synchronize()
    assert (Cm - ref).abs().max().item() <= 5e-3


def test_gemm_rejects_misaligned():
    dev = "cuda"
    A = _bf(torch.randn(16, 20, device=dev))
    B = _bf(torch.randn(16, 20, device=dev))
    Cm = torch.zeros(16, 16, device=dev)
    assert _lib().cc_gemm_op16_f32(OP[0], 0, 0, _p(A), 20, _p(B), 20, 16, 16, 20, _p(Cm), 16, None, 1, _st()) == -2


@pytest.mark.parametrize("rows,D", [(7, 64), (130, 768), (33, 1024), (5, 1600)])
def test_layernorm_fwd(rows, D):
    torch.manual_seed(rows + D)
    x = torch.randn(rows, D, device="cuda") * 2 + 0.3
    g = torch.randn(D, device="cuda")
    b = torch.randn(D, device="cuda")
    y = torch.empty(rows, D, dtype=OP[1], device="cuda")
    mean = torch.empty(rows, device="cuda")
    rstd = torch.empty(rows, device="cuda")
    assert _lib().cc_layernorm_fwd(OP[0], _p(x), _p(g), _p(b), _p(y), _p(mean), _p(rstd), rows, D, _st()) == 0
    ref = torch.nn.functional.layer_norm(x, (D,), g, b, 1e-5)
    torch.cuda.synchronize()
    assert (mean - x.mean(1)).abs().max() <= 1e-5
    assert (y.float() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item() + 1e-3


def _attn_ref(qkv, B, S, H, hd, causal):
    D = H * hd
    q, k, v = qkv.float().view(B, S, 3, H, hd).unbind(2)
    att = torch.einsum("bnhd,bmhd->bhnm", q, k) * hd ** -0.5
    if causal:
        att = att.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool, device=qkv.device), 1), float("-inf"))
    lse = torch.logsumexp(att, -1)
    p = att.softmax(-1)
    out = torch.einsum("bhnm,bmhd->bnhd", p, v).reshape(B, S, D)
    return out, lse


@pytest.mark.parametrize("B,S,H,hd,causal", [(3, 20, 8, 96, 0), (2, 50, 12, 64, 1), (2, 74, 4, 64, 1), (1, 7, 2, 8, 0), (2, 33, 2, 128, 1),
                                             (2, 64, 2, 64, 0), (1, 97, 2, 64, 1), (2, 32, 3, 96, 1), (2, 20, 4, 128, 0), (2, 40, 2, 96, 1),
                                             (1, 64, 2, 64, 1), (5, 17, 3, 64, 1)])
@pytest.mark.parametrize("mfma_bwd", [0, 1])
def test_attention_fwd_bwd(B, S, H, hd, causal, mfma_bwd):
    """hd in {64,96,128}: MFMA forward (and, with mfma_bwd, the MFMA backward: the one-pass kernel for S <= 64 (S <= 32 at hd 128), the
    dQ + dK/dV pair beyond); otherwise the LDS/VALU kernels.
    Reference: fp32 torch math on the same bf16 inputs; P is bf16-rounded before PV like the kernels do."""
    if not mfma_bwd and S > 80:
        pytest.skip("the LDS/VALU backward keeps whole S x S tiles in LDS (S <= ~80); longer sequences use the MFMA kernels")
    torch.manual_seed(S * 3 + hd)
    D = H * hd
    qkv = _bf(torch.randn(B * S, 3 * D, device="cuda"))
    out = torch.empty(B * S, D, dtype=OP[1], device="cuda")
    lse = torch.empty(B, H, S, device="cuda")
    assert _lib().cc_attention_fwd(OP[0], _p(qkv), B, S, H, hd, causal, _p(out), _p(lse), _st()) == 0
    qkv_r = qkv.float().requires_grad_(True)
    ref, lse_ref = _attn_ref(qkv_r, B, S, H, hd, causal)
    torch.cuda.synchronize()
    assert (lse - lse_ref).abs().max().item() <= 1e-4
    # bf16 output: <= 1 ulp of the value plus the bf16-P contribution
    tol = 2 ** -7 * ref.abs().clamp_min(0.25) + 4e-3
    assert ((out.float().view(B, S, D) - ref).abs() <= tol).all(), (out.float().view(B, S, D) - ref).abs().max().item()
    dout = _bf(torch.randn(B * S, D, device="cuda"))
    dqkv = torch.full_like(qkv, float("nan"))
    delta = torch.empty(B * H * S, device="cuda")
    rc = _lib().cc_attention_bwd(OP[0], _p(qkv), _p(dout), _p(out) if mfma_bwd else None, _p(lse), _p(delta) if mfma_bwd else None, B, S, H, hd, causal,
                                 _p(dqkv), _st())
    assert rc == 0
    ref.backward(dout.float().view(B, S, D))
    torch.cuda.synchronize()
    g = qkv_r.grad
    assert torch.isfinite(dqkv.float()).all()
    err = (dqkv.float() - g).abs().max().item()
    rel = ((dqkv.float() - g).norm() / g.norm()).item()
    assert rel <= 1e-2 and err <= 3e-2 * max(1.0, g.abs().max().item()), (rel, err)


def test_adamw_matches_torch_and_loss_scaling_rules():
    """cc_adamw_step == torch.optim.AdamW (model.py:73-77 defaults); with a loss scale the scaled gradients give the same update; a
    raised found_inf skips the step; cc_grad_nonfinite / cc_loss_scale_update follow torch.cuda.amp.GradScaler."""
    torch.manual_seed(0)
    n = 4096 + 64
    p = torch.randn(n, device="cuda")
    ref_p = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([ref_p], lr=3e-3)
    m = torch.zeros(n, device="cuda")
    v = torch.zeros(n, device="cuda")
    scale = torch.tensor([1024.0, 0.0], device="cuda")
    found = torch.zeros(1, device="cuda")
    l = _lib()
    for step in range(1, 4):
        g = torch.randn(n, device="cuda")
        ref_p.grad = g.clone()
        opt.step()
        if step == 2:      # scaled gradients + loss-scale pointer == unscaled gradients
            gs = g * 1024.0
            assert l.cc_adamw_step(_p(p), _p(gs), _p(m), _p(v), n, 3e-3, 0.9, 0.999, 1e-8, 0.01, step, 1.0, _p(scale), _p(found), _st()) == 0
        else:
            assert l.cc_adamw_step(_p(p), _p(g), _p(m), _p(v), n, 3e-3, 0.9, 0.999, 1e-8, 0.01, step, 1.0, None, None, _st()) == 0
    torch.cuda.synchronize()
    assert (p - ref_p.detach()).abs().max().item() <= 2e-6
    # overflow: found_inf is raised by an inf or a nan anywhere, the step is skipped, the scale halves and the flag clears
    before = (p.clone(), m.clone(), v.clone())
    for bad in (float("inf"), float("nan")):
        g = torch.randn(n, device="cuda")
        g[n - 3] = bad
        assert l.cc_grad_nonfinite(_p(g), n, _p(found), _st()) == 0
        assert float(found) == 1.0
        assert l.cc_adamw_step(_p(p), _p(g), _p(m), _p(v), n, 3e-3, 0.9, 0.999, 1e-8, 0.01, 4, 1.0, _p(scale), _p(found), _st()) == 0
        assert torch.equal(p, before[0]) and torch.equal(m, before[1]) and torch.equal(v, before[2])
        s0 = float(scale[0])
        assert l.cc_loss_scale_update(_p(scale), _p(found), 2.0, 0.5, 3, _st()) == 0
        assert float(scale[0]) == s0 * 0.5 and float(scale[1]) == 0.0 and float(found) == 0.0
    g = torch.randn(n, device="cuda")
    assert l.cc_grad_nonfinite(_p(g), n, _p(found), _st()) == 0 and float(found) == 0.0
    s0 = float(scale[0])
    for i in range(3):     # growth after `interval` consecutive good steps
        assert l.cc_loss_scale_update(_p(scale), _p(found), 2.0, 0.5, 3, _st()) == 0
        assert float(scale[0]) == (s0 if i < 2 else 2 * s0) and float(scale[1]) == (i + 1 if i < 2 else 0)


def test_adamw_step_cast_and_transpose_only_sync_equal_the_two_pass_form():
    """cc_adamw_step_cast = cc_adamw_step + the 16-bit cast of the updated parameters in the same pass (bit-identical to cc_cast_op16 of
    the result; untouched when found_inf skips the step), and cc_*_transpose_weights after it = cc_*_sync_weights: the whole operand
    arena (cast half and transposed half) is bit-identical either way."""
    from clipcap_amd import _lib as L
    l = _lib()
    torch.manual_seed(3)
    cfg = L.Gpt2Cfg(64, 4, 2, 97, 128, 16, OP[0])
    n = l.cc_gpt2_param_count(C.byref(cfg))
    mcfg = L.MapperCfg(E=32, D=64, P=3, L=2, H=4, N=2, Hm=128, W=1, use_pos=0, op_dtype=OP[0])
    nm = l.cc_mapper_param_count(C.byref(mcfg))
    for count, sync, transp, c in ((n, l.cc_gpt2_sync_weights, l.cc_gpt2_transpose_weights, cfg), (nm, l.cc_mapper_sync_weights, l.cc_mapper_transpose_weights, mcfg)):
        p0 = torch.randn(count, device="cuda")
        g = torch.randn(count, device="cuda")
        pa, pb = p0.clone(), p0.clone()
        ma, va, mb, vb = (torch.zeros(count, device="cuda") for _ in range(4))
        wa = torch.zeros(2 * count, dtype=OP[1], device="cuda")
        wb = torch.full((2 * count,), 7.0, dtype=OP[1], device="cuda")
        assert l.cc_adamw_step(_p(pa), _p(g), _p(ma), _p(va), count, 1e-2, 0.9, 0.999, 1e-8, 0.01, 1, 1.0, None, None, _st()) == 0
        assert sync(C.byref(c), _p(pa), _p(wa), _st()) == 0
        assert l.cc_adamw_step_cast(OP[0], _p(pb), _p(g), _p(mb), _p(vb), count, 1e-2, 0.9, 0.999, 1e-8, 0.01, 1, 1.0, None, None, _p(wb), _st()) == 0
        assert transp(C.byref(c), _p(wb), _st()) == 0
        torch.cuda.synchronize()
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
        assert torch.equal(wa[:count].view(torch.int16), wb[:count].view(torch.int16))
        # the transposed half: every GEMM weight's slot is rewritten by both forms; slots of 1-D tensors are never read or written
        gemm = wa[count:].float() != 0
        assert torch.equal(wa[count:][gemm].view(torch.int16), wb[count:][gemm].view(torch.int16)) and gemm.any()
        # a skipped step leaves the cast untouched
        found = torch.ones(1, device="cuda")
        scale = torch.tensor([8.0, 0.0], device="cuda")
        before = wb.clone()
        assert l.cc_adamw_step_cast(OP[0], _p(pb), _p(g), _p(mb), _p(vb), count, 1e-2, 0.9, 0.999, 1e-8, 0.01, 2, 1.0, _p(scale), _p(found), _p(wb), _st()) == 0
        torch.cuda.synchronize()
        assert torch.equal(pa, pb) and torch.equal(before.view(torch.int16), wb.view(torch.int16))
