"""GPU test of the data-parallel training step: two processes share the one MI355X of the test box (gloo backend on device
tensors, since RCCL needs one GPU per rank), each runs the real HIP step on its half of the global batch with the overlapped
gradient all-reduce; the result must equal the single-process step on the whole batch (global kept-token divisor)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.test_ddp_gloo import _free_port
from tests.util import load_golden, sd_of

pytestmark = pytest.mark.gpu


def _build(mode):
    from clipcap_amd.engine import ClipCapEngine, Gpt2Engine, MapperEngine
    g = load_golden(f"train_{mode}")
    E, D, P, L, H, N, n_head, n_layer, V, npos = [int(v) for v in g["cfg"]]
    sd = sd_of(g)
    me = MapperEngine(E, D, L, P, H, N, device="cuda")
    ge = Gpt2Engine(D, n_head, n_layer, V, npos, device="cuda")
    for k, v in me.views(me.arena.w32).items():
        v.copy_(sd["transformer_mapper." + k])
    for k, v in ge.views(ge.arena.w32).items():
        v.copy_(sd["language_model." + k])
    torch.manual_seed(5)
    tokens = torch.randint(1, V, (8, 8))
    tokens[1, 4:] = -1
    tokens[6, 2:] = -1
    embeds = torch.randn(8, E)
    return ClipCapEngine(me, ge, train_lm=(mode == "full")), tokens, embeds


def _worker(rank, world, port, mode, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from clipcap_amd.train.ddp import GradReducer, shard_batch
    eng, tokens, embeds = _build(mode)
    arenas = [eng.mapper.arena] + ([eng.gpt2.arena] if mode == "full" else [])
    red = GradReducer([a.grads() for a in arenas])
    tk, em = shard_batch(tokens, embeds, rank, world)
    red.begin()
    loss = eng.forward_backward(tk.cuda(), em.cuda(), reduce_stats=red.reduce_stats, on_grads_ready=red.on_grads_ready)
    red.finish()
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out, loss=float(loss), **{f"g{i}": a.g32.cpu().numpy() for i, a in enumerate(arenas)})
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["prefix_only", "full"])
def test_two_rank_step_equals_single_process(tmp_path, mode):
    out = str(tmp_path / "ddp.npz")
    mp.spawn(_worker, args=(2, _free_port(), mode, out), nprocs=2, join=True)
    res = np.load(out)
    eng, tokens, embeds = _build(mode)
    loss = eng.forward_backward(tokens.cuda(), embeds.cuda())
    torch.cuda.synchronize()
    assert abs(float(loss) - float(res["loss"])) <= 1e-5
    arenas = [eng.mapper.arena] + ([eng.gpt2.arena] if mode == "full" else [])
    for i, a in enumerate(arenas):
        ref = a.g32.cpu().numpy()
        rel = np.linalg.norm(res[f"g{i}"] - ref) / np.linalg.norm(ref)
        assert rel <= 2e-2, (i, rel)   # per-rank GEMMs see different M tiles / split-K slices: bf16-level agreement
