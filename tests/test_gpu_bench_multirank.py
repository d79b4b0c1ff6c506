"""bench.py's multi-rank flow (torch.distributed.run, one process per rank, barrier + max-over-ranks timing, GradReducer overlap) on a
single-GPU box: both ranks share cuda:0 and the collectives go over gloo (CC_BENCH_DEVICE / CC_BENCH_BACKEND test hooks; the driver's
real runs use one GPU per rank and RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_one_gpu():
    env = dict(os.environ, CC_BENCH_DEVICE="0", CC_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29517",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--batch", "64"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                 # rank 0 prints ONE JSON line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "weak" and j["config"]["global_batch"] == 128
    assert j["value"] > 0 and 0 < j["config"]["final_loss"] < 20
    assert j["roofline"]["bound"] == "mfma" and j["roofline"]["achieved"] > 0


def test_bench_spawns_its_own_ranks_and_refuses_a_short_run():
    """`python bench.py --gpus 2` without a launcher: bench.py starts the two ranks itself (same torchrun line the
    driver uses) and prints an n_gpus = 2 line; asked for more GPUs than are visible (and no test hook) it exits non-zero instead of
    printing a 1-rank line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CC_BENCH_DEVICE="0", CC_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--batch", "32",
           "--no-roofline-pass"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 64 and j["allreduce_exposed_ms"] is not None
    env.pop("CC_BENCH_DEVICE")
    import torch
    n = torch.cuda.device_count() + 1
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "refusing" in bad.stderr and not [l for l in bad.stdout.splitlines() if l.startswith("{")]
