"""Data-parallel glue: one process per GPU, gradients live in flat arenas, RCCL all-reduce over xGMI.

The reference has no explicit collective (Lightning + DeepSpeed do it implicitly, clipcap/train/train.py:77-85) and does not
shard its data (clipcap/train/dataloader.py:84-91), so the spec here is ours:
  * rank r takes rows [r*B/N, (r+1)*B/N) of every global batch (``shard_batch``);
  * the loss divisor is the GLOBAL number of kept targets: a 2-float [loss_sum, kept_count] all-reduce before backward
    (``reduce_stats``), so summing the ranks' gradients reproduces the single-process global-batch gradient exactly;
  * gradient arenas are all-reduced (SUM) in large buckets (``all_reduce``): xGMI is point-to-point, a ring moves
    2(N-1)/N of the payload over one link, so few large collectives beat many small ones.
This module is compute-agnostic (it only sees tensors), which is what lets tests/test_ddp_gloo.py run it on CPU with gloo.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous near-equal split of n rows; the first n % world ranks get one extra row."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(tokens: torch.Tensor, embeds: torch.Tensor, rank: int, world: int):
    lo, hi = shard_range(tokens.shape[0], rank, world)
    return tokens[lo:hi], embeds[lo:hi]


class GradReducer:
    """Bucketed SUM all-reduce of flat gradient arenas plus the loss-statistics reduction."""

    def __init__(self, flats: Sequence[torch.Tensor], bucket_bytes: int = 256 << 20, group=None):
        self.flats = list(flats)
        self.group = group
        self.buckets: List[torch.Tensor] = []
        for f in self.flats:
            assert f.dim() == 1 and f.is_contiguous()
            per = max(1, bucket_bytes // f.element_size())
            for lo in range(0, f.numel(), per):
                self.buckets.append(f[lo:lo + per])

    def reduce_stats(self, stats: torch.Tensor) -> None:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.group)

    def all_reduce(self) -> None:
        """Non-overlapped form: everything at once, after backward."""
        works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for b in self.buckets]
        for w in works:
            w.wait()

    # ---- overlapped form: slices are reduced as backward finishes them ------------------------------------------
    def begin(self) -> None:
        self._works = []
        self._covered = [0] * len(self.flats)

    def on_grads_ready(self, arena: int, lo: int, hi: int) -> None:
        """Backward has enqueued every kernel that writes flats[arena][lo:hi]: start its all-reduce now.  The collective is
        ordered after those kernels (ProcessGroupNCCL waits on the current stream) and runs on RCCL's own stream, i.e. under
        the backward kernels of the layers below."""
        if hi <= lo:
            return
        self._works.append(dist.all_reduce(self.flats[arena][lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._covered[arena] += hi - lo

    def finish(self) -> None:
        """Blocks the current stream until every slice is reduced; checks that the slices tiled each arena exactly once."""
        for w in self._works:
            w.wait()
        for f, c in zip(self.flats, self._covered):
            assert c == f.numel(), f"overlapped all-reduce covered {c} of {f.numel()} gradient elements"
        self._works = []
