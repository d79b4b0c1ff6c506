"""Input path: streams (tokens, embeddings) batches from the preprocessed dataset layout the reference trains on
(written by clipcap/preprocess/writer.py:49-75: ``embeddings/*.npy`` float arrays (n,E) or (n,W,E) + ``captions/*.parquet``
with a ``caption`` column) — the contract of clipcap/train/dataloader.py:11-66:

    batch = (tokens int64 (B, max_token_length) right-padded with -1 / truncated, embeds float32 (B, E))

The reference delegates to the un-vendored ``embedding_reader`` package; this is a native reader: ``np.load(mmap_mode="r")``
per shard, pyarrow for captions, contiguous per-rank row ranges (the reference does not shard at all, dataloader.py:84-91),
pinned host staging and an async H2D copy on a side stream so the next batch uploads while the current step runs.
"""
from __future__ import annotations

import glob
import os
from typing import Iterator, List, Optional, Tuple

import numpy as np
import torch


class ShardIndex:
    """Row-addressable view over the sorted embedding / caption shard pairs."""

    def __init__(self, data_path: str):
        self.emb_files = sorted(glob.glob(os.path.join(data_path, "embeddings", "*.npy")))
        self.cap_files = sorted(glob.glob(os.path.join(data_path, "captions", "*.parquet")))
        if not self.emb_files or len(self.emb_files) != len(self.cap_files):
            raise FileNotFoundError(f"{data_path}: need matching embeddings/*.npy and captions/*.parquet shards "
                                    f"(found {len(self.emb_files)} / {len(self.cap_files)})")
        self.arrays = [np.load(f, mmap_mode="r") for f in self.emb_files]
        self.counts = [a.shape[0] for a in self.arrays]
        self.starts = np.concatenate([[0], np.cumsum(self.counts)])
        self.count = int(self.starts[-1])
        self.dimension = int(self.arrays[0].shape[-1])
        self.sample_shape = tuple(self.arrays[0].shape[1:])
        self._caps: dict = {}

    def _captions(self, shard: int) -> List[str]:
        if shard not in self._caps:
            import pyarrow.parquet as pq
            self._caps = {shard: pq.read_table(self.cap_files[shard], columns=["caption"]).column("caption").to_pylist()}
        return self._caps[shard]

    def rows(self, lo: int, hi: int) -> Tuple[np.ndarray, List[str]]:
        embs, caps = [], []
        s = int(np.searchsorted(self.starts, lo, side="right") - 1)
        while lo < hi:
            a, b = lo - self.starts[s], min(hi, self.starts[s + 1]) - self.starts[s]
            embs.append(np.array(self.arrays[s][a:b], dtype=np.float32))   # copy out of the read-only mmap
            caps += self._captions(s)[a:b]
            lo = int(self.starts[s] + b)
            s += 1
        return (embs[0] if len(embs) == 1 else np.concatenate(embs)), caps


def pad_tokens(ids: List[int], max_token_length: int) -> np.ndarray:
    """dataloader.py:41-50: right-pad with -1 to max_token_length, or truncate."""
    out = np.full(max_token_length, -1, dtype=np.int64)
    n = min(len(ids), max_token_length)
    out[:n] = ids[:n]
    return out


class EmbedDataset(torch.utils.data.IterableDataset):
    """Same constructor arguments as the reference's EmbedDataset (dataloader.py:16-17) + rank/world for sharding."""

    def __init__(self, data_path: str = "./dataset/", language_model: str = "gpt2-xl", batch_size: int = 256,
                 reader_max_piece_size: int = 50, reader_parallel_pieces: int = 10, max_token_length: int = 64, tokenizer=None,
                 rank: int = 0, world_size: int = 1) -> None:
        super().__init__()
        if tokenizer is None:
            from clipcap_amd.model.model import get_tokenizer
            tokenizer = get_tokenizer(language_model)
        self.tokenizer = tokenizer
        self.batch_size = batch_size
        self.max_token_length = max_token_length
        self.index = ShardIndex(data_path)
        self.encoder_embedding_size = self.index.dimension
        self.rank, self.world_size = rank, world_size

    def __len__(self) -> int:
        """Optimizer steps per epoch — identical on every rank.  A trailing global batch with fewer rows than ranks is dropped on
        ALL ranks: otherwise some ranks would get zero rows and skip the step while the others enter its collectives."""
        gb = self.batch_size * self.world_size
        full, tail = divmod(self.index.count, gb)
        return full + (1 if tail >= self.world_size else 0)

    def encode(self, captions: List[str]) -> np.ndarray:
        enc = self.tokenizer.batch_encode_plus(captions)["input_ids"] if hasattr(self.tokenizer, "batch_encode_plus") \
            else [self.tokenizer.encode(c) for c in captions]
        return np.stack([pad_tokens(ids, self.max_token_length) for ids in enc])

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        from clipcap_amd.train.ddp import shard_range
        gb = self.batch_size * self.world_size
        for lo in range(0, self.index.count, gb):
            hi = min(self.index.count, lo + gb)
            if hi - lo < self.world_size:          # fewer rows than ranks: no rank takes this step (see __len__)
                break
            a, b = shard_range(hi - lo, self.rank, self.world_size)
            emb, caps = self.index.rows(lo + a, lo + b)
            yield torch.from_numpy(self.encode(caps)), torch.from_numpy(np.ascontiguousarray(emb))


def get_dataloader(data_path: str = "./dataset/", language_model: str = "gpt2-xl", batch_size: int = 256, tokenizer=None, rank: int = 0,
                   world_size: int = 1):
    """Returns (iterable of batches, encoder_embedding_size) like dataloader.py:69-92."""
    ds = EmbedDataset(data_path=data_path, language_model=language_model, batch_size=batch_size, tokenizer=tokenizer, rank=rank,
                      world_size=world_size)
    return ds, ds.encoder_embedding_size


def trim_padding(tokens: torch.Tensor, multiple: int = 8) -> torch.Tensor:
    """Drops the all-padding tail columns of a right-padded (-1) token batch (kept length rounded up to `multiple`).
    Exact: a dropped column has no kept target (model.py:103-109), and under the causal mask no kept row attends to it, so
    loss and gradients are unchanged while GPT-2 runs on L + longest_caption rows instead of L + max_token_length."""
    if tokens.numel() == 0:
        return tokens
    lengths = tokens.ge(0).sum(dim=1)
    keep = int(lengths.max())
    keep = min(tokens.shape[1], max(multiple, (keep + multiple - 1) // multiple * multiple))
    return tokens[:, :keep].contiguous()


class DevicePrefetcher:
    """Uploads batch i+1 (pinned staging + non_blocking copy on a side stream) while batch i is being consumed; trims the
    all-padding tail of the token batch on the host first (``trim_padding``)."""

    def __init__(self, it, device, trim: bool = True):
        self.it = iter(it)
        self.trim = trim
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self._next = self._load()

    def _load(self):
        try:
            tokens, emb = next(self.it)
        except StopIteration:
            return None
        if self.trim:
            tokens = trim_padding(tokens)
        if self.stream is None:
            return tokens, emb
        with torch.cuda.stream(self.stream):
            return tokens.pin_memory().to(self.device, non_blocking=True), emb.pin_memory().to(self.device, non_blocking=True)

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        cur = self._next
        if self.stream is not None:
            compute = torch.cuda.current_stream(self.device)
            compute.wait_stream(self.stream)
            # the batch was allocated on the side stream: tell the caching allocator that the compute stream uses it, so that its
            # block is not handed to a later side-stream upload while this step's kernels are still queued
            for t in cur:
                t.record_stream(compute)
        self._next = self._load()
        return cur
