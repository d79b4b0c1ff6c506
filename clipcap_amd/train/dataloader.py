"""Input path: streams (tokens, embeddings) batches from the preprocessed dataset layout the reference trains on
(written by clipcap/preprocess/writer.py:49-75: ``embeddings/*.npy`` float arrays (n,E) or (n,W,E) + ``captions/*.parquet``
with a ``caption`` column) — the contract of clipcap/train/dataloader.py:11-66:

    batch = (tokens int64 (B, max_token_length) right-padded with -1 / truncated, embeds float32 (B, E))

The reference delegates to the un-vendored ``embedding_reader`` package, which reads ``parallel_pieces`` pieces of at most
``max_piece_size`` MB concurrently and hands batches out in order (dataloader.py:32-37, 53-63).  This is a native reader with the same two
knobs: ``np.load(mmap_mode="r")`` per shard, pyarrow for captions, contiguous per-rank row ranges (the reference does not shard at all,
dataloader.py:84-91); a bounded background pipeline of ``reader_parallel_pieces`` workers, each reading + tokenising + padding one PIECE
(consecutive global batches, at most ``reader_max_piece_size`` MB of embeddings) at a time, delivered strictly in order — worker PROCESSES
(spawned once, reused across epochs) whenever the tokenizer can be pickled: the tokenizers library does not tokenise concurrently from
several threads of one interpreter (measured: 4 threads = 4 sequential calls), and 256 captions cost ~6 ms of it per 10 ms step; threads
otherwise; persistent
pinned staging buffers and an async H2D copy on a side stream (DevicePrefetcher), so that the training thread does no shard I/O and no
tokenisation between launches (bench.py --mode e2e measures what is left).
"""
from __future__ import annotations

import glob
import multiprocessing
import os
import pickle
import threading
from collections import OrderedDict, deque
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor
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
        self._caps: "OrderedDict[int, List[str]]" = OrderedDict()      # a few decoded caption shards (worker threads read concurrently)
        self._caps_lock = threading.Lock()

    def _captions(self, shard: int) -> List[str]:
        with self._caps_lock:
            caps = self._caps.get(shard)
            if caps is not None:
                self._caps.move_to_end(shard)
                return caps
        import pyarrow.parquet as pq
        caps = pq.read_table(self.cap_files[shard], columns=["caption"]).column("caption").to_pylist()
        with self._caps_lock:
            self._caps[shard] = caps
            while len(self._caps) > 4:
                self._caps.popitem(last=False)
        return caps

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


_WORKER_DS = None      # reader worker process: its own sequential EmbedDataset (own mmaps, own tokenizer copy)


def _reader_worker_init(kwargs: dict) -> None:
    global _WORKER_DS
    try:
        torch.set_num_threads(1)
    except RuntimeError:
        pass
    _WORKER_DS = EmbedDataset(**kwargs)


def _reader_worker_piece(lo: int, hi: int):
    """One piece in a worker process: numpy arrays travel back (cheaper to pickle than tensors)."""
    return [(t.numpy(), e.numpy()) for t, e in _WORKER_DS.load_piece(lo, hi)]


class EmbedDataset(torch.utils.data.IterableDataset):
    """Same constructor arguments as the reference's EmbedDataset (dataloader.py:16-17) + rank/world for sharding."""

    def __init__(self, data_path: str = "./dataset/", language_model: str = "gpt2-xl", batch_size: int = 256,
                 reader_max_piece_size: int = 50, reader_parallel_pieces: int = 10, max_token_length: int = 64, tokenizer=None,
                 rank: int = 0, world_size: int = 1, reader_backend: str = "auto") -> None:
        """``reader_backend``: "process" / "thread" / "auto" (processes when the tokenizer pickles, see the module docstring)."""
        super().__init__()
        self.data_path, self.language_model = data_path, language_model
        self.reader_backend = reader_backend
        self._pool = None
        if tokenizer is None:
            from clipcap_amd.model.model import get_tokenizer
            tokenizer = get_tokenizer(language_model)
        self.tokenizer = tokenizer
        self.batch_size = batch_size
        self.max_token_length = max_token_length
        self.reader_max_piece_size = reader_max_piece_size          # MB of embeddings read (and tokenised) by one worker at a time
        self.reader_parallel_pieces = reader_parallel_pieces        # pieces in flight = worker threads (<= 0: everything on the caller's thread)
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
        # dataloader.py:41-50 for the whole piece at once (right-pad with -1 / truncate) without a Python-level loop over tokens: the workers
        # share the interpreter lock with the training thread, so everything here is one C-level pass (chain -> fromiter -> fancy index)
        import itertools
        L, n = self.max_token_length, len(enc)
        lens = np.fromiter(map(len, enc), dtype=np.int64, count=n)
        flat = np.fromiter(itertools.chain.from_iterable(enc), dtype=np.int64, count=int(lens.sum()))
        rows = np.repeat(np.arange(n), lens)
        cols = np.arange(flat.size) - np.repeat(np.cumsum(lens) - lens, lens)
        keep = cols < L
        out = np.full((n, L), -1, dtype=np.int64)
        out[rows[keep], cols[keep]] = flat[keep]
        return out

    MAX_BATCHES_PER_PIECE = 8      # a piece is also bounded in batches, so that the first batch of an epoch is not behind 50 MB of tokenisation

    def pieces(self) -> List[Tuple[int, int]]:
        """Global row ranges [lo, hi) of the pieces of one epoch: whole global batches (the last one may be partial), at most
        reader_max_piece_size MB of embeddings and MAX_BATCHES_PER_PIECE batches each.  Identical on every rank."""
        gb = self.batch_size * self.world_size
        row_bytes = 4 * int(np.prod(self.index.sample_shape))
        k = max(1, min(self.MAX_BATCHES_PER_PIECE, int(self.reader_max_piece_size * (1 << 20)) // max(1, gb * row_bytes)))
        out = []
        for lo in range(0, self.index.count, gb * k):
            out.append((lo, min(self.index.count, lo + gb * k)))
        return out

    def load_piece(self, lo: int, hi: int) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        """This rank's batches of the global rows [lo, hi): shard reads, ONE tokeniser call for the whole piece, padding."""
        from clipcap_amd.train.ddp import shard_range
        gb = self.batch_size * self.world_size
        embs, caps, counts = [], [], []
        for b0 in range(lo, hi, gb):
            b1 = min(hi, b0 + gb)
            if b1 - b0 < self.world_size:          # fewer rows than ranks: no rank takes this step (see __len__)
                break
            a, b = shard_range(b1 - b0, self.rank, self.world_size)
            e, c = self.index.rows(b0 + a, b0 + b)
            embs.append(e)
            caps += c
            counts.append(b - a)
        if not counts:
            return []
        toks = self.encode(caps)
        out, at = [], 0
        for e, n in zip(embs, counts):
            out.append((torch.from_numpy(toks[at:at + n]), torch.from_numpy(np.ascontiguousarray(e))))
            at += n
        return out

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        pieces = self.pieces()
        par = int(self.reader_parallel_pieces)
        if par > 0:      # one pool per rank: share the host's cores between the ranks of the node (8 ranks x 10 workers = 80 torch-importing processes otherwise)
            par = max(1, min(par, (os.cpu_count() or par) // max(1, int(self.world_size))))
        if par <= 0:
            for lo, hi in pieces:
                yield from self.load_piece(lo, hi)
            return
        # bounded, ordered: at most `par` pieces are being read / tokenised ahead of the consumer; results are handed out in piece order
        ex, own = self._executor(par)
        conv = (lambda bs: [(torch.from_numpy(t), torch.from_numpy(e)) for t, e in bs]) if isinstance(ex, ProcessPoolExecutor) else (lambda bs: bs)
        fn = _reader_worker_piece if isinstance(ex, ProcessPoolExecutor) else self.load_piece
        pending: deque = deque()
        it = iter(pieces)
        try:
            for _ in range(par):
                nxt = next(it, None)
                if nxt is None:
                    break
                pending.append(ex.submit(fn, *nxt))
            first = True
            while pending:
                fut = pending.popleft()
                try:
                    # a worker process that cannot start (an un-importable __main__, a tokenizer that unpickles badly) must not hang the run
                    batches = conv(fut.result(timeout=300 if first and not own else None))
                except Exception as e:       # BrokenProcessPool, TimeoutError, an exception raised in the worker
                    if own:
                        raise
                    import warnings
                    warnings.warn(f"clipcap_amd reader: worker processes failed ({type(e).__name__}: {e}); reading on threads instead")
                    self.close()
                    self.reader_backend = "thread"
                    done = len(pieces) - len(pending) - 1 - sum(1 for _ in it)          # pieces already delivered
                    for lo, hi in pieces[done:]:
                        yield from self.load_piece(lo, hi)
                    return
                first = False
                nxt = next(it, None)
                if nxt is not None:
                    pending.append(ex.submit(fn, *nxt))
                yield from batches
        finally:
            for f in pending:
                f.cancel()
            if own:
                ex.shutdown(wait=True, cancel_futures=True)

    def _executor(self, par: int):
        """(executor, caller_owns_it).  The process pool is created once and kept for the following epochs (a spawned worker imports torch:
        seconds, once); a thread pool is per epoch."""
        backend = self.reader_backend
        if backend == "auto":
            try:
                pickle.dumps(self.tokenizer)
                backend = "process"
            except Exception:      # a tokenizer that cannot travel (a local class, an object holding handles): threads
                backend = "thread"
        if backend == "thread":
            return ThreadPoolExecutor(max_workers=par, thread_name_prefix="clipcap-reader"), True
        if self._pool is None or self._pool[1] != par:
            self.close()
            kwargs = dict(data_path=self.data_path, language_model=self.language_model, batch_size=self.batch_size,
                          reader_max_piece_size=self.reader_max_piece_size, reader_parallel_pieces=0, max_token_length=self.max_token_length,
                          tokenizer=self.tokenizer, rank=self.rank, world_size=self.world_size)
            # spawn, not fork: the parent holds a HIP runtime and (in train()) RCCL threads
            ex = ProcessPoolExecutor(max_workers=par, mp_context=multiprocessing.get_context("spawn"), initializer=_reader_worker_init, initargs=(kwargs,))
            self._pool = (ex, par)
        return self._pool[0], False

    def close(self) -> None:
        if self._pool is not None:
            self._pool[0].shutdown(wait=False, cancel_futures=True)
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def get_dataloader(data_path: str = "./dataset/", language_model: str = "gpt2-xl", batch_size: int = 256, tokenizer=None, rank: int = 0,
                   world_size: int = 1, reader_max_piece_size: int = 50, reader_parallel_pieces: int = 10, max_token_length: int = 64):
    """Returns (iterable of batches, encoder_embedding_size) like dataloader.py:69-92."""
    ds = EmbedDataset(data_path=data_path, language_model=language_model, batch_size=batch_size, tokenizer=tokenizer, rank=rank,
                      world_size=world_size, reader_max_piece_size=reader_max_piece_size, reader_parallel_pieces=reader_parallel_pieces,
                      max_token_length=max_token_length)
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
    """Uploads batch i+1 while batch i is being consumed: the all-padding tail of the token batch is trimmed on the host (``trim_padding``),
    the batch is copied into one of ``depth`` PERSISTENT pinned staging buffers (a fresh ``pin_memory()`` per batch is a hipHostMalloc per
    step) and sent with a non-blocking copy on a side stream; a staging buffer is reused only after the copy that read it has finished."""

    def __init__(self, it, device, trim: bool = True, depth: int = 2):
        self.it = iter(it)
        self.trim = trim
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self._slots = [None] * max(2, depth)       # (pinned token bytes, pinned embedding bytes, event of the last copy out of them)
        self._n = 0
        self._next = self._load()

    def _staged(self, slot, which, t: torch.Tensor) -> torch.Tensor:
        need = t.numel() * t.element_size()
        buf = slot[which]
        if buf is None or buf.numel() < need:
            buf = torch.empty(max(need, 1), dtype=torch.uint8).pin_memory()
            slot[which] = buf
        view = buf[:need].view(t.dtype).view(t.shape)
        # a plain single-threaded memcpy: torch's copy_ splits half a megabyte over the intra-op thread pool, and waking that pool from the
        # training loop cost 5-25 ms per step on a 256-core host (tools/e2e_host_profile.py) against 5 us for the copy itself
        np.copyto(view.numpy(), t.numpy())
        return view

    def _load(self):
        try:
            tokens, emb = next(self.it)
        except StopIteration:
            return None
        if self.trim:
            tokens = trim_padding(tokens)
        if self.stream is None:
            return tokens, emb
        i = self._n % len(self._slots)
        self._n += 1
        if self._slots[i] is None:
            self._slots[i] = [None, None, torch.cuda.Event()]
        slot = self._slots[i]
        if slot[0] is not None:
            slot[2].synchronize()                   # the H2D copy that last read these staging buffers is done
        pt, pe = self._staged(slot, 0, tokens), self._staged(slot, 1, emb)
        with torch.cuda.stream(self.stream):
            out = pt.to(self.device, non_blocking=True), pe.to(self.device, non_blocking=True)
            slot[2].record(self.stream)
        return out

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
