"""world_size-2 gloo test (CPU) of the data-parallel spec in clipcap_amd/train/ddp.py: rank-sharded batches + global
kept-token divisor + SUM all-reduce of flat gradient arenas == the single-process gradient on the global batch.
The per-rank compute is the CPU oracle (the checker standing in for the HIP step, which needs a GPU)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import clipcap_oracle as O
from tests.util import load_golden, sd_of


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup():
    g = load_golden("train_prefix_only")
    E, D, P, L, H, N, n_head, n_layer, V, npos = [int(v) for v in g["cfg"]]
    cfg = dict(projection_length=P, prefix_length=L, heads=H, layers=N, n_head=n_head, n_layer=n_layer)
    sd = sd_of(g)
    sd.pop("language_model.lm_head.weight", None)
    torch.manual_seed(3)
    Bg = 6
    tokens = torch.randint(1, V, (Bg, 8))
    tokens[0, 3:] = -1          # ranks see different kept-token counts
    tokens[4, 6:] = -1
    tokens[5, 1] = 0
    embeds = torch.randn(Bg, E)
    return cfg, sd, tokens, embeds


def _grads_flat(sd, names, tokens, embeds, cfg, denom):
    for k in names:
        sd[k].requires_grad_(True)
        sd[k].grad = None
    t = tokens.clone()
    loss = O.clipcap_loss(sd, t, embeds, cfg=cfg, denom=denom)
    loss.backward()
    return torch.cat([sd[k].grad.reshape(-1) for k in names]), float(loss.detach())


def _worker(rank, world, port, out, wire="fp32"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from clipcap_amd.train.ddp import GradReducer, shard_batch
    cfg, sd, tokens, embeds = _setup()
    names = sorted(k for k in sd if k.startswith("transformer_mapper."))
    tk, em = shard_batch(tokens, embeds, rank, world)
    kept = float(((tk > 0)).sum())
    stats = torch.tensor([0.0, kept])
    flat = torch.zeros(sum(sd[k].numel() for k in names))
    red = GradReducer([flat], bucket_bytes=1 << 16, wire_dtype=torch.bfloat16 if wire == "bf16" else torch.float32)
    assert len(red.buckets) > 1
    red.reduce_stats(stats)                      # global kept-token count
    g, local_sum = _grads_flat(sd, names, tk, em, cfg, denom=float(stats[1]))
    flat.copy_(g)
    if rank == 0:      # exercise both forms: they must agree
        pass
    n = flat.numel()
    red.begin()        # overlapped form: slices handed over from the top of the arena to the bottom, like backward does
    cuts = [n, n - n // 3, n // 4, 0]
    for hi, lo in zip(cuts, cuts[1:]):
        red.on_grads_ready(0, lo, hi)
    red.finish()
    loss_stats = torch.tensor([local_sum])
    dist.all_reduce(loss_stats)
    if rank == 0:
        np.save(out, np.concatenate([flat.numpy(), loss_stats.numpy()]))
    dist.destroy_process_group()


def test_two_rank_gradients_equal_single_process(tmp_path):
    out = str(tmp_path / "ddp.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = np.load(out)
    cfg, sd, tokens, embeds = _setup()
    names = sorted(k for k in sd if k.startswith("transformer_mapper."))
    ref, ref_loss = _grads_flat(sd, names, tokens, embeds, cfg, denom=None)   # reference semantics: mean over kept targets
    assert abs(res[-1] - ref_loss) <= 1e-5
    err = np.abs(res[:-1] - ref.numpy()).max()
    assert err <= 1e-6 * max(1.0, float(ref.abs().max())), err


def test_two_rank_gradients_bf16_on_the_wire(tmp_path):
    """wire_dtype=bf16: slices are cast to bf16, summed in bf16 across the ranks and widened back into the fp32 arena — the
    single-process gradient to bf16 precision (two roundings: the cast and the 2-rank sum)."""
    out = str(tmp_path / "ddp16.npy")
    mp.spawn(_worker, args=(2, _free_port(), out, "bf16"), nprocs=2, join=True)
    res = np.load(out)
    cfg, sd, tokens, embeds = _setup()
    names = sorted(k for k in sd if k.startswith("transformer_mapper."))
    ref, ref_loss = _grads_flat(sd, names, tokens, embeds, cfg, denom=None)
    assert abs(res[-1] - ref_loss) <= 1e-5
    rel = np.linalg.norm(res[:-1] - ref.numpy()) / np.linalg.norm(ref.numpy())
    assert 1e-5 < rel <= 8e-3, rel          # really went through bf16, and no worse than two bf16 roundings


def test_shard_range_covers_batch():
    from clipcap_amd.train.ddp import shard_range
    for n in (1, 7, 8, 256, 257):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _train_worker(rank, world, port, out, wire, steps):
    """`steps` optimizer steps of the tiny frozen-LM model on 2 ranks: rank-sharded batch, global kept-token divisor, GradReducer with
    the given wire dtype, AdamW + linear schedule (oracle arithmetic standing in for the HIP step)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from clipcap_amd.train.ddp import GradReducer, shard_batch
    cfg, sd, tokens, embeds = _setup()
    names = sorted(k for k in sd if k.startswith("transformer_mapper."))
    sizes = [sd[k].numel() for k in names]
    flat = torch.zeros(sum(sizes))
    red = GradReducer([flat], bucket_bytes=1 << 16, wire_dtype=torch.bfloat16 if wire == "bf16" else torch.float32)
    m = {k: torch.zeros_like(sd[k]) for k in names}
    v = {k: torch.zeros_like(sd[k]) for k in names}
    losses = []
    for step in range(steps):
        tk, em = shard_batch(tokens, embeds, rank, world)                # the same global batch every step (the model can fit it)
        stats = torch.tensor([0.0, float((tk > 0).sum())])
        red.reduce_stats(stats)
        g, local_sum = _grads_flat(sd, names, tk, em, cfg, denom=float(stats[1]))
        flat.copy_(g)
        red.all_reduce()
        ls = torch.tensor([local_sum])
        dist.all_reduce(ls)
        losses.append(float(ls))
        lr = 3e-3 * O.linear_schedule_factor(step, 2, steps + 4)
        with torch.no_grad():
            off = 0
            for k, n in zip(names, sizes):
                pn, m[k], v[k] = O.adamw_step(sd[k].detach(), flat[off:off + n].view_as(sd[k]), m[k], v[k], step + 1, lr)
                sd[k] = pn.detach()
                off += n
    if rank == 0:
        np.save(out, np.array(losses))
    dist.destroy_process_group()


def test_twenty_step_loss_trajectory_bf16_wire_follows_fp32_wire(tmp_path):
    """The bf16 gradient wire (default for frozen-LM runs: half the all-reduce bytes for the 41.7 M mapper gradients that only the
    short mapper backward can hide) against the fp32 wire over 20 AdamW steps of the same 2-rank run: the loss goes down and the two
    trajectories stay together to well under the step-to-step change."""
    traj = {}
    for wire in ("fp32", "bf16"):
        out = str(tmp_path / f"traj_{wire}.npy")
        mp.spawn(_train_worker, args=(2, _free_port(), out, wire, 20), nprocs=2, join=True)
        traj[wire] = np.load(out)
    a, b = traj["fp32"], traj["bf16"]
    assert a[-5:].mean() < a[:5].mean() - 0.05                  # it trains
    assert np.abs(a - b).max() <= 2e-3 * np.abs(a).max(), np.abs(a - b).max()
    assert np.abs(a - b).max() <= 0.1 * np.abs(np.diff(a)).mean() + 1e-4 or np.abs(a - b).max() <= 1e-3


def test_bench_eight_rank_dry_run_every_rank_leaves_before_rank_zero_extras():
    """no 8-GPU node exists for this build, so the first 8-rank run of bench.py must not be the first time its control flow
    executes.  `bench.py --gpus 8 --dry-run` walks the real flow of that file (warm-up, timed regions with max-over-ranks, the
    no-all-reduce pass, both event-bracketed passes, teardown) over gloo with a stub step; rank 0 then spends seconds alone in its
    single-rank extras.  Every rank must have left the process group BEFORE that, the run must end cleanly and print ONE 8-rank line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CC_BENCH_DRY_EXTRAS_S="3.0")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    t0 = __import__("time").time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "3", "--warmup", "1", "--regions", "3"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    left = __import__("re").findall(r"\[bench rank (\d+)\] left the process group", out.stderr)      # (ranks share the pipe: lines may run together)
    assert sorted(int(r) for r in left) == list(range(8)), left
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["dry_run"] is True and d["config"]["global_batch"] == 8 * 256 and d["config"]["parallelism"] == "dp8"
    assert d["timed_regions"]["regions"] == 3 and len(d["timed_regions"]["ms_per_step_each"]) == 3
    assert d["timed_regions"]["ms_per_step_min"] <= d["ms_per_step"] <= d["timed_regions"]["ms_per_step_max"]
    assert "cpu_baseline" in d and "allreduce_exposed_ms" in d and d["collective_backend"] == "gloo"
    print(f"8-rank dry run: {__import__('time').time() - t0:.1f} s")


def _adamw_ref(w, g, m, v, lr, t, b1=0.9, b2=0.999, eps=1e-8, wd=0.01):
    """torch.optim.AdamW's update, in place, on flat tensors (what cc_adamw_step computes on the device)."""
    w.mul_(1 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    w.addcdiv_(m / (1 - b1 ** t), (v / (1 - b2 ** t)).sqrt_().add_(eps), value=-lr)


def _zero_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from clipcap_amd.engine import _Arena
    from clipcap_amd.train.ddp import ZeroShard
    n = 1003                                              # not a multiple of the world size or of the 8-element slice granule
    gen = torch.Generator().manual_seed(11)
    a = _Arena(n, "cpu")
    a.w32.copy_(torch.randn(n, generator=gen))
    if rank == 0:                                        # a resumed run: full moments present before sharding is configured
        a.m, a.v = torch.zeros(n), torch.zeros(n)
    ZeroShard(rank, world).apply([a])
    r, ranges, gather = a.zero
    lo, hi = ranges[rank]
    assert ranges[0][0] == 0 and ranges[-1][1] == n and all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
    assert all(lo_ % 8 == 0 for lo_, _ in ranges)
    if a.m is None:
        a.m, a.v = torch.zeros(hi - lo), torch.zeros(hi - lo)
    assert a.m.numel() == hi - lo
    for t in range(1, 4):
        g = torch.randn(n, generator=gen)                # the all-reduced gradient: the same on every rank
        _adamw_ref(a.w32[lo:hi], g[lo:hi], a.m, a.v, 1e-2, t)
        gather(a.w32, ranges)
    m, v = a.full_moments()                              # collective
    if rank == 0:
        np.savez(out, w=a.w32.numpy(), m=m.numpy(), v=v.numpy())
    dist.destroy_process_group()


def test_sharded_optimizer_state_three_ranks(tmp_path):
    """ddp.ZeroShard + _Arena.shard_optimizer_state / full_moments on 3 gloo ranks (uneven slices): stepping the own slice and broadcasting
    the owners' slices reproduces the replicated update exactly; the gathered moments are the replicated moments."""
    from clipcap_amd.train.ddp import zero_stage
    out = str(tmp_path / "zero.npz")
    mp.spawn(_zero_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    res = np.load(out)
    n = 1003
    gen = torch.Generator().manual_seed(11)
    w, m, v = torch.randn(n, generator=gen), torch.zeros(n), torch.zeros(n)
    for t in range(1, 4):
        _adamw_ref(w, torch.randn(n, generator=gen), m, v, 1e-2, t)
    assert np.array_equal(res["w"], w.numpy()) and np.array_equal(res["m"], m.numpy()) and np.array_equal(res["v"], v.numpy())
    assert [zero_stage(s) for s in (None, "", "deepspeed_stage_1", "deepspeed_stage_2_offload", "deepspeed_stage_3", "2", "deepspeed", "zero1", "ddp")] == [0, 0, 1, 2, 2, 2, 2, 1, 0]


def _zero2_worker(rank, world, port, out, wire):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from clipcap_amd.engine import _Arena
    from clipcap_amd.train.ddp import GradReducer, ZeroShard
    sizes = (1003, 517)                                   # two arenas (mapper, language model), neither a multiple of world or of 8
    arenas = []
    for i, n in enumerate(sizes):
        a = _Arena(n, "cpu")
        a.w32.copy_(torch.randn(n, generator=torch.Generator().manual_seed(20 + i)))
        arenas.append(a)
    red = GradReducer([a.grads() for a in arenas], wire_dtype=torch.bfloat16 if wire == "bf16" else torch.float32)
    owners = ZeroShard(rank, world).apply(arenas)
    red.set_owners(owners, rank)
    flag_seen = []
    for t in range(1, 4):
        for i, a in enumerate(arenas):                    # this rank's local gradient
            a.grads().copy_(torch.randn(a.n, generator=torch.Generator().manual_seed(1000 * t + 10 * i + rank)))
        red.begin()                                       # slices arrive the way backward hands them out: top first, cut anywhere
        n0, n1 = sizes
        red.on_grads_ready(1, 300, n1)
        red.on_grads_ready(1, 0, 300)
        red.on_grads_ready(0, 641, n0)
        red.on_grads_ready(0, 0, 641)
        red.finish()
        flag = torch.tensor([1.0 if (t == 2 and rank == 1) else 0.0])      # one rank's slice overflowed: every rank must see it
        red.reduce_flag(flag)
        flag_seen.append(float(flag))
        for a in arenas:
            r, ranges, gather = a.zero
            lo, hi = ranges[r]
            if a.m is None:
                a.m, a.v = torch.zeros(hi - lo), torch.zeros(hi - lo)
            if flag_seen[-1] == 0.0:
                _adamw_ref(a.w32[lo:hi], a.grads()[lo:hi], a.m, a.v, 1e-2, t)
            gather(a.w32, ranges)
    if rank == 0:
        np.savez(out, w0=arenas[0].w32.numpy(), w1=arenas[1].w32.numpy(), flags=np.array(flag_seen))
    dist.destroy_process_group()


def _zero2_expected(world, wire):
    sizes = (1003, 517)
    ws = [torch.randn(n, generator=torch.Generator().manual_seed(20 + i)) for i, n in enumerate(sizes)]
    ms = [torch.zeros(n) for n in sizes]
    vs = [torch.zeros(n) for n in sizes]
    for t in range(1, 4):
        for i, n in enumerate(sizes):
            parts = [torch.randn(n, generator=torch.Generator().manual_seed(1000 * t + 10 * i + r)) for r in range(world)]
            if wire == "bf16":
                acc = parts[0].to(torch.bfloat16)
                for q in parts[1:]:
                    acc = (acc.float() + q.to(torch.bfloat16).float()).to(torch.bfloat16)       # gloo's bf16 SUM: add in fp32, round per hop
                g = acc.float()
            else:
                g = parts[0].clone()
                for q in parts[1:]:
                    g += q
            if t != 2:
                _adamw_ref(ws[i], g, ms[i], vs[i], 1e-2, t)
    return ws


def test_partitioned_gradients_two_ranks_fp32_and_bf16_wire(tmp_path):
    """ZeRO stage 2 (--deepspeed-strategy deepspeed_stage_2): GradReducer.set_owners reduces every gradient slice onto its owner only, the
    owner steps its slice and the slices are broadcast back — after three steps (one of them skipped everywhere because ONE rank flagged an
    overflow) the parameters equal the replicated all-reduce + AdamW result bit for bit, for both wire formats."""
    for wire in ("fp32", "bf16"):
        out = str(tmp_path / f"zero2_{wire}.npz")
        mp.spawn(_zero2_worker, args=(2, _free_port(), out, wire), nprocs=2, join=True)
        res = np.load(out)
        ws = _zero2_expected(2, wire)
        assert list(res["flags"]) == [0.0, 1.0, 0.0]
        assert np.array_equal(res["w0"], ws[0].numpy()) and np.array_equal(res["w1"], ws[1].numpy()), wire


def test_partitioned_gradients_three_ranks(tmp_path):
    """Same with three ranks (uneven owner ranges; slices that straddle two owner boundaries): fp32 wire, sums within 1 ulp-level tolerance
    (the reduction order of three addends is the backend's)."""
    out = str(tmp_path / "zero2_3.npz")
    mp.spawn(_zero2_worker, args=(3, _free_port(), out, "fp32"), nprocs=3, join=True)
    res = np.load(out)
    ws = _zero2_expected(3, "fp32")
    assert list(res["flags"]) == [0.0, 1.0, 0.0]
    assert np.allclose(res["w0"], ws[0].numpy(), rtol=0, atol=2e-6) and np.allclose(res["w1"], ws[1].numpy(), rtol=0, atol=2e-6)

