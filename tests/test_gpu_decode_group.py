"""Beam-group form of the KV-cached attention step (cc_decode_fwd_g, clipcap_amd/csrc/decode.hip::k_decode_attn_group): the rows of a
beam group read every distinct (cache row, position) of their ancestry tables once.  `group` is a performance hint — the logits must be
those of the per-row kernel (cc_decode_fwd_p) and of the full re-forward the reference does (inference/base.py:80-121) for ANY
ancestry table: shared prefixes, partially shared histories, rows that share nothing, rows that name other groups' cache rows.

(The persistent-launch forms of the layer stack — decode_pk.hip, decode_xt.hip — are lab-build code: tests/lab_decode_pk.py, tests/lab_decode_xt.py,
run by tests/test_gpu_lab.py (marker `lab`) against libclipcap_hip_lab.so; the weight-image GEMM form: tests/lab_decode_image.py.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lockstep(lm, S, G, L0, steps, tol, seed=0, cross_group=False):
    """Two sessions fed the same inputs and reorders — one stepping with group=1, one with group=G — plus the full re-forward."""
    from clipcap_amd.engine import DecodeSession
    ge = lm.engine
    D = ge.dims["D"]
    gen = torch.Generator(device="cuda").manual_seed(seed)
    R = S * G
    pref = torch.randn(S, L0, D, generator=gen, device="cuda") * 0.5
    base = torch.arange(S, device="cuda", dtype=torch.int32).repeat_interleave(G)
    sess = [DecodeSession(ge, S, L0 + steps + 1) for _ in range(2)]
    for s in sess:
        s.forward(pref)
    sess = [s.expand(base, R) for s in sess]
    assert int(sess[1].row_map[:, 0].unique().numel()) == S        # the fanned-out rows name ONE copy of their prefix
    hist = pref[base.long()]                                       # (R, t, D): every row's own input history, for the re-forward
    worst = 0.0
    for t in range(steps):
        x = torch.randn(R, 1, D, generator=gen, device="cuda") * 0.5
        la = sess[0].forward(x, group=1).clone()
        lb = sess[1].forward(x, group=G).clone()
        sess[1].check()                                            # the persistent layer launch finished every hand-off
        hist = torch.cat((hist, x), dim=1)
        scale = max(1.0, la.abs().max().item())
        d = (la - lb).abs().max().item() / scale
        worst = max(worst, d)
        assert d <= tol, (t, d)
        if t in (0, steps - 1):                                   # the reference's semantics: re-forward of the whole history
            chk = torch.arange(0, R, max(1, R // 6), device="cuda")
            full = ge.logits(hist[chk])[:, -1]
            assert ((lb[chk] - full).abs().max().item() / scale) <= 4 * tol, t
        # a beam step's reorder: inside the group (as beam search does), sometimes every row from one ancestor, sometimes identity
        if t % 3 == 2:
            loc = torch.zeros(R, dtype=torch.int64, device="cuda")
        elif t % 3 == 1:
            loc = torch.arange(R, device="cuda") % G
        else:
            loc = torch.randint(0, G, (R,), generator=gen, device="cuda")
        src = (torch.arange(R, device="cuda") // G) * G + loc
        if cross_group and t == 1:                                # not a beam search, but legal for the API: rows adopt other groups' histories
            src = torch.randint(0, R, (R,), generator=gen, device="cuda")
        for s in sess:
            s.reorder(src)
        hist = hist[src]
    return worst


@pytest.mark.parametrize("G", [2, 3, 5, 8])
def test_group_attention_equals_per_row_kernel_tiny(G):
    from tests.test_gpu_api import _model_from_train_fixture
    m, _ = _model_from_train_fixture()
    w = _lockstep(m.language_model, 3, G, 4, 7, 2e-3, seed=G, cross_group=True)
    print(f"group {G}: worst |group - per-row| / scale = {w:.2e}")


@pytest.mark.parametrize("precision", [None, 16, 32])
def test_group_attention_medium_width_320_rows(precision):
    """GPT-2-medium width (16 heads of 64), 2 layers, 64 prefixes x beam 5 — the configs[4] geometry — in the three operand modes."""
    from tests.test_gpu_configs import _medium_lm
    lm, _ = _medium_lm(2, precision=precision)
    tol = 1e-4 if precision == 32 else 4e-3
    w = _lockstep(lm, 64, 5, 10, 6, tol, seed=11)
    print(f"precision {precision}: worst |group - per-row| / scale = {w:.2e}")


def test_group_hint_is_ignored_where_it_does_not_apply():
    """group that does not divide the row count, group > 8 and multi-position calls fall back to the per-row kernel."""
    from clipcap_amd.engine import DecodeSession
    from tests.test_gpu_api import _model_from_train_fixture
    m, _ = _model_from_train_fixture()
    ge = m.language_model.engine
    torch.manual_seed(3)
    x = torch.randn(9, 6, 64, device="cuda") * 0.5
    ref = ge.logits(x)
    for grp in (4, 9, 3):
        s = DecodeSession(ge, 9, 8)
        l0 = s.forward(x[:, :5], group=grp).clone()               # Tnew = 5: always per-row
        l1 = s.forward(x[:, 5:6], group=grp)
        assert (l0 - ref[:, 4]).abs().max().item() <= 2e-3 and (l1 - ref[:, 5]).abs().max().item() <= 2e-3, grp


