"""The lab build (clipcap_amd/libclipcap_hip_lab.so = `make -C clipcap_amd/csrc lab`: the product library plus the experiment kernels and the
environment A/B switches, clipcap_amd/csrc/lab_env.h) is exercised in subprocesses, so that what HISTORY.md 4.5 says "passes either way" is
covered: the persistent decode-layer launch (cc_decode_mode bit 1), the XCD-team decode engine (bit 2), the weight-image decode GEMMs (bit 3),
the fp32-MFMA attention kernels (CC_ATTN_F32MFMA=1), the VALU attention fallbacks (CC_ATTN_VALU=1).  The product library carries none of them
(and ignores the environment): tests/test_api_surface.py checks that on the CPU.

Marker `lab`, NOT `gpu`: `pytest -m gpu` (the product's GPU suite) does not select these; run them with `pytest -m lab tests/test_gpu_lab.py` on a
GPU box (tools/profile_round.sh does)."""
import os
import subprocess
import sys

import pytest

import torch

pytestmark = [pytest.mark.lab, pytest.mark.skipif(not torch.cuda.is_available(), reason="lab-build experiments need a GPU")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB = os.path.join(ROOT, "clipcap_amd", "libclipcap_hip_lab.so")


@pytest.mark.parametrize("name,args,env", [
    ("persistent decode-layer launch", ["tests/lab_decode_pk.py"], {}),
    ("XCD-team decode engine", ["tests/lab_decode_xt.py"], {}),
    ("weight-image decode GEMMs", ["tests/lab_decode_image.py"], {}),
    ("fp32-MFMA attention (split-bf16 mode)", ["tests/test_gpu_x3.py", "-k", "attention_kernels or dropout or autograd or windowed"], {"CC_ATTN_F32MFMA": "1"}),
    # (S = 97 is beyond the LDS-tile kernels' whole-sequence tile: CC_ERR_SHAPE is their documented answer)
    ("VALU attention fallbacks", ["tests/test_gpu_kernels.py", "-k", "attention and not 1-97-2-64"], {"CC_ATTN_VALU": "1"}),
])
def test_lab_build(name, args, env):
    assert os.path.exists(LAB), "build the lab library: make -C clipcap_amd/csrc lab (__graft_entry__.build() does)"
    e = dict(os.environ, CLIPCAP_HIP_LIB="lab", **env)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", *args], cwd=ROOT, env=e, capture_output=True, text=True, timeout=3000)
    tail = (r.stdout or "")[-3000:] + (r.stderr or "")[-1500:]
    assert r.returncode == 0, f"{name}: {tail}"
    assert " passed" in r.stdout, tail
