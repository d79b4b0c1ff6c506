"""LAB BUILD ONLY (libclipcap_hip_lab.so, run by tests/test_gpu_lab.py with CLIPCAP_HIP_LIB=lab): cc_decode_mode bit 1 — the whole layer stack of a
group step as ONE persistent launch whose workgroups hand activations over through arrival counters (clipcap_amd/csrc/decode_pk.hip).  An A/B
switch (measured 2x slower than the per-op launches, HISTORY.md 4.5), held to the bars of tests/test_gpu_decode_group.py, plus
cc_decode_ws_check (no hand-off gave up).  Reference semantics: the full re-forward per generated token, clipcap/inference/base.py:80-121."""
import pytest
import torch  # noqa: F401

from tests.test_gpu_decode_group import _lockstep

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision,S,G,NL", [(None, 64, 5, 3), (16, 64, 5, 2), (None, 13, 3, 2), (None, 32, 8, 2), (None, 7, 2, 2), (32, 64, 5, 2)])
def test_persistent_layer_launch_equals_per_op_launches(precision, S, G, NL):
    """cc_decode_mode bit 1: one persistent launch per position (decode_pk.hip) against the per-row launches and the full re-forward —
    320 rows, ragged row tiles (39 / 14 rows), group widths 2 / 3 / 5 / 8; split-bf16 operands keep the per-op path (the flag is then a
    no-op and the result must not change)."""
    from clipcap_amd import _lib
    from tests.test_gpu_configs import _medium_lm
    lm, _ = _medium_lm(NL, precision=precision)
    old = _lib.lib().cc_decode_mode(3)
    try:
        w = _lockstep(lm, S, G, 10, 6, 1e-4 if precision == 32 else 4e-3, seed=100 + S)
    finally:
        _lib.lib().cc_decode_mode(old)
    print(f"persistent launch, precision {precision}, {S} x {G} rows, {NL} layers: worst |persistent - per-row launches| / scale = {w:.2e}")
