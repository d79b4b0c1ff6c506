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
    """mode "prefix_only" / "full" (bf16 operands) or "prefix_only_fp16" / "full_fp16" (fp16 operands + loss scale)."""
    from clipcap_amd.engine import ClipCapEngine, Gpt2Engine, MapperEngine
    precision = 16 if mode.endswith("_fp16") else None
    mode = mode.replace("_fp16", "")
    g = load_golden(f"train_{mode}")
    E, D, P, L, H, N, n_head, n_layer, V, npos = [int(v) for v in g["cfg"]]
    sd = sd_of(g)
    me = MapperEngine(E, D, L, P, H, N, device="cuda", precision=precision)
    ge = Gpt2Engine(D, n_head, n_layer, V, npos, device="cuda", precision=precision)
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
    arenas = eng.arenas()
    red = GradReducer([a.grads() for a in arenas])
    tk, em = shard_batch(tokens, embeds, rank, world)
    red.begin()
    loss = eng.forward_backward(tk.cuda(), em.cuda(), reduce_stats=red.reduce_stats, on_grads_ready=red.on_grads_ready)
    red.finish()
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out, loss=float(loss), **{f"g{i}": a.g32.cpu().numpy() for i, a in enumerate(arenas)})
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["prefix_only", "full", "full_fp16"])
def test_two_rank_step_equals_single_process(tmp_path, mode):
    out = str(tmp_path / "ddp.npz")
    mp.spawn(_worker, args=(2, _free_port(), mode, out), nprocs=2, join=True)
    res = np.load(out)
    eng, tokens, embeds = _build(mode)
    loss = eng.forward_backward(tokens.cuda(), embeds.cuda())
    torch.cuda.synchronize()
    assert abs(float(loss) - float(res["loss"])) <= 1e-5
    for i, a in enumerate(eng.arenas()):
        ref = a.g32.cpu().numpy()       # fp16 operands: both sides carry the same initial loss scale
        rel = np.linalg.norm(res[f"g{i}"] - ref) / np.linalg.norm(ref)
        assert rel <= 2e-2, (i, rel)   # per-rank GEMMs see different M tiles / split-K slices: bf16-level agreement


def test_c_abi_rccl_communicator_single_rank_and_overlapped_reducer():
    """cc_comm_create / cc_allreduce_bucket / cc_comm_destroy on the one GPU of the test box: a 1-rank RCCL communicator
    (all-reduce = identity) driven through GradReducer's overlapped path gives exactly the gradients of the plain step.  The same
    code with nranks > 1 is what a multi-GPU binder runs."""
    from clipcap_amd.train.ddp import CAbiComm, GradReducer
    comm = CAbiComm(1, 0, CAbiComm.unique_id(), "cuda:0")
    try:
        for dt in (torch.float32, torch.bfloat16, torch.float16):
            t = torch.randn(4099, device="cuda").to(dt)
            ref = t.clone()
            comm.all_reduce_(t)
            torch.cuda.synchronize()
            assert torch.equal(t, ref)
            comm.reduce_(t, 0)                     # cc_reduce_bucket = ncclReduce onto rank 0
            torch.cuda.synchronize()
            assert torch.equal(t, ref)
        eng, tokens, embeds = _build("full")
        loss0 = eng.forward_backward(tokens.cuda(), embeds.cuda())
        torch.cuda.synchronize()
        ref = [a.g32.clone() for a in eng.arenas()]
        eng.zero_grad()
        red = GradReducer([a.grads() for a in eng.arenas()], comm=comm)
        red.begin()
        loss1 = eng.forward_backward(tokens.cuda(), embeds.cuda(), reduce_stats=red.reduce_stats, on_grads_ready=red.on_grads_ready)
        red.finish()
        torch.cuda.synchronize()
        assert float(loss0) == float(loss1)
        for a, r in zip(eng.arenas(), ref):
            assert torch.allclose(a.g32, r, rtol=1e-4, atol=1e-7)
        # sharded optimizer state over the same communicator (cc_broadcast_bucket = ncclBroadcast): with one rank the own slice is the
        # whole arena and the step must equal the replicated step
        from clipcap_amd.train.ddp import ZeroShard
        w_before = [a.w32.clone() for a in eng.arenas()]
        eng.optimizer_step(1e-3, 1, weight_decay=0.01)
        w_rep = [a.w32.clone() for a in eng.arenas()]
        for a, w in zip(eng.arenas(), w_before):
            a.w32.copy_(w)
            a.m = a.v = None
        ZeroShard(0, 1, comm=comm).apply(eng.arenas())
        eng.optimizer_step(1e-3, 1, weight_decay=0.01)
        torch.cuda.synchronize()
        for a, w in zip(eng.arenas(), w_rep):
            assert torch.equal(a.w32, w)
            m, v = a.full_moments()
            assert m.numel() == a.n and float(m.abs().max()) > 0
        # ZeRO stage 2 over the same communicator: every slice reduced onto its owner (cc_reduce_bucket), fp32 and bf16 wire; with one
        # rank the owner is this rank and the gradients must come out as the plain step's
        eng.zero_grad()
        eng.forward_backward(tokens.cuda(), embeds.cuda())             # the plain step at the updated parameters
        ref = [a.g32.clone() for a in eng.arenas()]
        for wire in (torch.float32, torch.bfloat16):
            eng.zero_grad()
            red2 = GradReducer([a.grads() for a in eng.arenas()], comm=comm, wire_dtype=wire)
            red2.set_owners([a.zero[1] for a in eng.arenas()], 0)
            red2.begin()
            eng.forward_backward(tokens.cuda(), embeds.cuda(), reduce_stats=red2.reduce_stats, on_grads_ready=red2.on_grads_ready)
            red2.finish()
            flag = torch.zeros(1, device="cuda")
            red2.reduce_flag(flag)
            torch.cuda.synchronize()
            assert float(flag) == 0.0
            for a, r in zip(eng.arenas(), ref):
                if wire == torch.float32:
                    assert torch.allclose(a.g32, r, rtol=1e-4, atol=1e-7)
                else:
                    assert torch.allclose(a.g32, r.to(torch.bfloat16).float(), rtol=2e-2, atol=1e-6)
    finally:
        comm.close()


def _nccl_worker(out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)        # "nccl" IS RCCL on ROCm
    from clipcap_amd.train.ddp import GradReducer
    eng, tokens, embeds = _build("prefix_only")
    red = GradReducer([a.grads() for a in eng.arenas()])
    red.begin()
    loss = eng.forward_backward(tokens.cuda(), embeds.cuda(), reduce_stats=red.reduce_stats, on_grads_ready=red.on_grads_ready)
    red.finish()
    torch.cuda.synchronize()
    g_allreduce = eng.mapper.arena.g32.clone()
    # ZeRO stage 2 on the same process group: dist.reduce onto the owner (one rank: this one)
    from clipcap_amd.train.ddp import ZeroShard
    owners = ZeroShard(0, 1).apply(eng.arenas())
    red.set_owners(owners, 0)
    eng.zero_grad()
    red.begin()
    eng.forward_backward(tokens.cuda(), embeds.cuda(), reduce_stats=red.reduce_stats, on_grads_ready=red.on_grads_ready)
    red.finish()
    torch.cuda.synchronize()
    assert torch.allclose(eng.mapper.arena.g32, g_allreduce, rtol=1e-4, atol=1e-7)
    np.savez(out, loss=float(loss), g=g_allreduce.cpu().numpy())
    dist.destroy_process_group()


def test_torch_distributed_rccl_backend_executes():
    """The production collective path (torch.distributed backend "nccl" = RCCL, device_id-bound process group, async all-reduce of
    arena slices overlapped with backward) actually runs: one rank on the one GPU here; with >= 2 GPUs the 2-rank variant below."""
    import tempfile
    out = os.path.join(tempfile.mkdtemp(), "nccl1.npz")
    mp.spawn(_nccl_worker_entry, args=(out,), nprocs=1, join=True)
    res = np.load(out)
    eng, tokens, embeds = _build("prefix_only")
    loss = eng.forward_backward(tokens.cuda(), embeds.cuda())
    torch.cuda.synchronize()
    assert abs(float(loss) - float(res["loss"])) <= 1e-6
    assert np.allclose(res["g"], eng.mapper.arena.g32.cpu().numpy(), rtol=1e-4, atol=1e-7)


def _nccl_worker_entry(rank, out):
    _nccl_worker(out)


def _nccl2_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from clipcap_amd.train.ddp import GradReducer, shard_batch
    eng, tokens, embeds = _build("full")
    for e in (eng.mapper, eng.gpt2):
        e.to(dev)
    red = GradReducer([a.grads() for a in eng.arenas()])
    tk, em = shard_batch(tokens, embeds, rank, world)
    red.begin()
    loss = eng.forward_backward(tk.to(dev), em.to(dev), reduce_stats=red.reduce_stats, on_grads_ready=red.on_grads_ready)
    red.finish()
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out, loss=float(loss), **{f"g{i}": a.g32.cpu().numpy() for i, a in enumerate(eng.arenas())})
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL requires one device per rank")
def test_two_rank_rccl_step_equals_single_process(tmp_path):
    out = str(tmp_path / "rccl2.npz")
    mp.spawn(_nccl2_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = np.load(out)
    eng, tokens, embeds = _build("full")
    loss = eng.forward_backward(tokens.cuda(), embeds.cuda())
    torch.cuda.synchronize()
    assert abs(float(loss) - float(res["loss"])) <= 1e-5
    for i, a in enumerate(eng.arenas()):
        ref = a.g32.cpu().numpy()
        assert np.linalg.norm(res[f"g{i}"] - ref) / np.linalg.norm(ref) <= 2e-2


def test_gradient_wire_kernels_equal_torch_casts_at_any_alignment():
    """cc_grad_wire_pack / _unpack (the bf16 wire of GradReducer on the GPU): bit-identical to torch's fp32 -> bf16 -> fp32 casts for odd
    lengths, slices that start at odd element offsets, denormals, infinities and NaN."""
    from clipcap_amd.train.ddp import _wire_cast
    torch.manual_seed(0)
    base = torch.randn(100_003, device="cuda") * torch.logspace(-30, 30, 100_003, device="cuda")
    base[17] = float("inf"); base[18] = float("-inf"); base[19] = float("nan"); base[20] = 1e-41; base[21] = 0.0
    stage = torch.empty(100_003, dtype=torch.bfloat16, device="cuda")
    back = torch.full((100_003,), 7.0, device="cuda")
    for lo, hi in ((0, 100_003), (1, 9), (3, 100_000), (5, 6), (8, 4104), (2, 2)):
        _wire_cast(base[lo:hi], stage[lo:hi])
        want = base[lo:hi].to(torch.bfloat16)
        assert torch.equal(stage[lo:hi].view(torch.int16), want.view(torch.int16)) or \
            torch.equal(torch.nan_to_num(stage[lo:hi].float(), nan=123.0), torch.nan_to_num(want.float(), nan=123.0)), (lo, hi)
        _wire_cast(stage[lo:hi], back[lo:hi])
        assert torch.equal(torch.nan_to_num(back[lo:hi], nan=123.0), torch.nan_to_num(want.float(), nan=123.0)), (lo, hi)
    assert float(back[100_002]) == float(base[100_002].to(torch.bfloat16).float())


def _zero_worker(rank, world, port, sharded, out, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from clipcap_amd.train.ddp import GradReducer, ZeroShard, shard_batch
    eng, tokens, embeds = _build(mode)
    arenas = eng.arenas()
    red = GradReducer([a.grads() for a in arenas])
    if sharded:
        ZeroShard(rank, world).apply(arenas)
    tk, em = shard_batch(tokens, embeds, rank, world)
    for step in range(1, 4):
        eng.zero_grad()
        red.begin()
        eng.forward_backward(tk.cuda(), em.cuda(), reduce_stats=red.reduce_stats, on_grads_ready=red.on_grads_ready)
        red.finish()
        eng.optimizer_step(1e-3, step, weight_decay=0.01)
    if sharded:
        lo, hi = arenas[0].zero[1][rank]
        assert arenas[0].m.numel() == hi - lo < arenas[0].n            # this rank holds its own slice of the moments only
    moments = [a.full_moments() for a in arenas]                      # collective when sharded
    logits = eng.gpt2.logits(torch.randn(2, 5, eng.gpt2.dims["D"], generator=torch.Generator().manual_seed(1)).cuda())   # the operand copies were refreshed
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out, logits=logits.cpu().numpy(), **{f"w{i}": a.w32.cpu().numpy() for i, a in enumerate(arenas)},
                 **{f"m{i}": m.cpu().numpy() for i, (m, _) in enumerate(moments)}, **{f"v{i}": v.cpu().numpy() for i, (_, v) in enumerate(moments)})
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["prefix_only", "full"])
def test_sharded_optimizer_state_steps_equal_replicated_steps(tmp_path, mode):
    """--deepspeed-strategy stage >= 1 (ddp.ZeroShard): each of 2 ranks keeps and steps half of the AdamW moments; after 3 steps the
    parameters, the gathered moments and the logits through the refreshed operand copies equal the replicated-state run.  Two runs of
    the same step differ in the last bits (LayerNorm parameter gradients and the token-embedding gradient are summed with fp32 atomics),
    so the bar is 1e-6 of each tensor's scale; the exact equality of the sharded update is tests/test_ddp_gloo.py's (3 ranks, CPU)."""
    outs = []
    for sharded in (False, True):
        out = str(tmp_path / f"zero{int(sharded)}.npz")
        mp.spawn(_zero_worker, args=(2, _free_port(), sharded, out, mode), nprocs=2, join=True)
        outs.append(np.load(out))
    a, b = outs
    for k in a.files:
        tol = 1e-6 * max(1e-30, np.abs(a[k]).max()) if k[0] in "mv" else 1e-6 * max(1.0, np.abs(a[k]).max()) + (2e-2 if k == "logits" else 0.0)
        d = np.abs(a[k] - b[k])
        # Adam divides by sqrt(v): an element whose gradient is at the rounding noise of the atomics may step differently in two runs
        # (at most lr per step) — a handful of such elements are tolerated in the parameters, none in the moments' scale
        assert (d > tol).mean() <= (1e-4 if k[0] == "w" else 0.0) and d.max() <= (4e-3 if k[0] == "w" else tol), (k, d.max(), tol)
    assert np.abs(a["m0"]).max() > 0
