"""LAB BUILD ONLY (libclipcap_hip_lab.so, run by tests/test_gpu_lab.py with CLIPCAP_HIP_LIB=lab): cc_decode_mode bit 3 — the K-split decode GEMMs read
their weight operand global -> VGPR from the fragment-ordered image of cc_decode_image (include/clipcap_hip_lab.h).  Bit-identical to the product
path, 1-3 % slower in the real decode chain (HISTORY.md): an A/B arm, not the product."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", [None, 16])
def test_weight_image_path_is_bit_identical(precision):
    """cc_decode_mode bit 3 (an A/B switch, default off): the K-split decode GEMMs (c_attn, mlp.c_proj at 320 rows) read their weight operand
    global -> VGPR from the fragment-ordered image of cc_decode_image (gemm.hip.h::gemm_nt_s64kwb_kernel) instead of staging it through LDS —
    same MFMA sequence, same summation order, so the logits must be IDENTICAL; no image exists while the bit is off, and a changed weight
    rebuilds it (a stale image would serve the old weights)."""
    from clipcap_amd import _lib
    from clipcap_amd.engine import DecodeSession
    from tests.test_gpu_configs import _medium_lm
    lm, _ = _medium_lm(2, precision=precision)
    ge = lm.engine
    l = _lib.lib()
    D = ge.dims["D"]
    gen = torch.Generator(device="cuda").manual_seed(5)
    S, G, L0 = 64, 5, 10
    pref = torch.randn(S, L0, D, generator=gen, device="cuda") * 0.5
    xs = [torch.randn(S * G, 1, D, generator=gen, device="cuda") * 0.5 for _ in range(3)]
    base = torch.arange(S, device="cuda", dtype=torch.int32).repeat_interleave(G)

    def run(on):
        old = l.cc_decode_mode((l.cc_decode_mode(-1) | 8) if on else (l.cc_decode_mode(-1) & ~8))
        try:
            s = DecodeSession(ge, S, L0 + 4)
            s.forward(pref)
            s = s.expand(base, S * G)
            out = [s.forward(x, group=G).clone() for x in xs]
            imgs = ge.decode_images()
            return out, imgs[0]
        finally:
            l.cc_decode_mode(old)

    off, img_off = run(False)
    on, img_on = run(True)
    assert img_off is None and img_on is not None
    for a, b in zip(off, on):
        assert torch.equal(a, b)
    with torch.no_grad():                                         # a weight change must reach the image
        dict(lm.named_parameters())["transformer.h.1.mlp.c_proj.weight"].mul_(0.5)
    off2, _ = run(False)
    on2, _ = run(True)
    assert not torch.equal(off2[0], off[0])
    for a, b in zip(off2, on2):
        assert torch.equal(a, b)
