"""LAB BUILD ONLY (libclipcap_hip_lab.so, run by tests/test_gpu_lab.py with CLIPCAP_HIP_LIB=lab).  XCD-team decode engine (cc_decode_fwd_x with a weight image, clipcap_amd/csrc/decode_xt.hip): every XCD runs the whole GPT-2 layer stack
of a generated position for its own captions, the weights stream into registers from a fragment-ordered image.  Replaces the per-token
full re-forward of the reference (clipcap/inference/base.py:80-121); must give the logits of the launch-per-op path (cc_decode_fwd_p) and
of the re-forward for ANY beam ancestry — the bars are those of tests/test_gpu_decode_group.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lm(D, n_layer, precision=None, seed=4402, npos=128, V=50257):
    from tests import seeded
    from clipcap_amd.model.gpt2 import GPT2LM
    gsd = seeded.state_dict(seeded.gpt2_shapes(D, n_layer, V, npos), seed)
    gsd["transformer.wte.weight"] = gsd["transformer.wte.weight"] * 2.0
    lm = GPT2LM(n_embd=D, n_layer=n_layer, n_head=D // 64, vocab_size=V, n_positions=npos, precision=precision)
    lm.load_state_dict({k: torch.from_numpy(v) for k, v in gsd.items()}, strict=False)
    return lm.to("cuda")


def _run(lm, S, G, steps=6, tol=4e-3, seed=0, expect_xt=True):
    from clipcap_amd import _lib
    from tests.test_gpu_decode_group import _lockstep
    l = _lib.lib()
    old = l.cc_decode_mode(l.cc_decode_mode(-1) | 4)          # the engine is an A/B switch (off by default): on for these tests
    try:
        w = _lockstep(lm, S, G, 10, steps, tol, seed=seed)
        assert l.cc_decode_last_path() == (2 if expect_xt else 0), "which path served the last group step"
    finally:
        l.cc_decode_mode(old)
    return w


@pytest.mark.parametrize("precision,S,G,NL", [(None, 64, 5, 3), (16, 64, 5, 2), (None, 13, 3, 2), (None, 32, 6, 2), (None, 7, 2, 2), (None, 9, 5, 2), (None, 40, 5, 2)])
def test_xcd_team_engine_equals_per_op_launches_medium_width(precision, S, G, NL):
    """D = 1024 (GPT-2-medium width): the configs[4] geometry (64 x 5 = 40 rows per XCD), ragged teams (13 captions over 8 XCDs: one team
    with a single caption, one with none), fewer captions than XCDs (7), group widths 2 / 3 / 5 / 6, fp16 operands."""
    lm = _lm(1024, NL, precision)
    w = _run(lm, S, G, seed=200 + S)
    print(f"XCD-team engine, precision {precision}, {S} x {G} rows, {NL} layers: worst |engine - per-op launches| / scale = {w:.2e}")


def test_xcd_team_engine_width_512():
    lm = _lm(512, 3)
    w = _run(lm, 24, 4, seed=5)
    print(f"XCD-team engine, D = 512: worst |engine - per-op launches| / scale = {w:.2e}")


@pytest.mark.parametrize("precision,D,S,G", [(32, 1024, 64, 5), (None, 768, 16, 5), (None, 1024, 80, 5)])
def test_geometries_the_engine_does_not_cover_keep_the_per_op_path(precision, D, S, G):
    """split-bf16 operands, GPT-2-small width, more than 48 rows per XCD: cc_decode_fwd_x is then exactly cc_decode_fwd_g."""
    lm = _lm(D, 2, precision)
    w = _run(lm, S, G, steps=4, tol=1e-4 if precision == 32 else 4e-3, seed=7, expect_xt=False)
    print(f"fallback, precision {precision}, D {D}, {S} x {G}: {w:.2e}")


def test_engine_off_switch_and_image_rebuild_after_a_weight_change():
    """cc_decode_mode bit 2 off -> no image is built and the per-op path runs; a changed parameter rebuilds the image (the engine must not
    decode with stale weights)."""
    from clipcap_amd import _lib
    from clipcap_amd.engine import DecodeSession
    l = _lib.lib()
    lm = _lm(1024, 2)
    ge = lm.engine
    torch.manual_seed(1)
    pref = torch.randn(8, 6, 1024, device="cuda") * 0.5
    x = torch.randn(40, 1, 1024, device="cuda") * 0.5
    base = torch.arange(8, device="cuda", dtype=torch.int32).repeat_interleave(5)

    def step():
        s = DecodeSession(ge, 8, 16)
        s.forward(pref)
        s = s.expand(base, 40)
        out = s.forward(x, group=5).clone()
        s.check()
        return out

    def with_mode(on):
        old = l.cc_decode_mode((l.cc_decode_mode(-1) | 4) if on else (l.cc_decode_mode(-1) & ~4))
        try:
            out = step()
            assert l.cc_decode_last_path() == (2 if on else 0)
            return out
        finally:
            l.cc_decode_mode(old)

    b = with_mode(False)
    assert (getattr(ge, "_imgs", None) or (None, None))[1] is None, "no engine image is built while the engine is off"
    a = with_mode(True)
    scale = max(1.0, b.abs().max().item())
    assert (a - b).abs().max().item() / scale <= 4e-3
    with torch.no_grad():
        dict(lm.named_parameters())["transformer.h.1.mlp.c_proj.weight"].mul_(0.5)
    c = with_mode(True)
    d = with_mode(False)
    assert (c - d).abs().max().item() / scale <= 4e-3 and (c - a).abs().max().item() / scale > 1e-2
