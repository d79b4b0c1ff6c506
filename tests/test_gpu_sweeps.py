"""A few seconds of each randomised parity sweep (tools/fuzz_*.py) with a fixed seed, so that the sweeps themselves stay runnable and a
regression in the paths they cover shows up in the suite; the long runs quoted in DESIGN.md 2 are done by hand."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,seconds", [("fuzz_kernels.py", 5), ("fuzz_decode_steps.py", 6), ("fuzz_model.py", 8), ("fuzz_beam_search.py", 6)])
def test_sweep_runs_clean(tool, seconds):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(seconds), "20260928"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert " 0 failures" in r.stdout, r.stdout[-500:]
