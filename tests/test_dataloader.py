"""The input path (clipcap_amd/train/dataloader.py; reference clipcap/train/dataloader.py:11-66 over the un-vendored embedding_reader): the background
piece pipeline (reader_parallel_pieces workers, pieces of at most reader_max_piece_size MB) must hand out exactly the batches the sequential
reader does, in the same order, on every rank — whatever the worker count."""
import os

import numpy as np
import pytest
import torch

pa = pytest.importorskip("pyarrow")
import pyarrow.parquet as pq  # noqa: E402


class _Tok:
    """caption 'cap <id> w w ...' -> [id, 7, 7, ...]: the row's identity survives tokenisation."""

    def encode(self, c):
        p = c.split()
        return [int(p[1])] + [7] * (len(p) - 2)


def _dataset(tmp, counts=(700, 333, 512), E=16, seed=0):
    os.makedirs(os.path.join(tmp, "embeddings"))
    os.makedirs(os.path.join(tmp, "captions"))
    rng = np.random.default_rng(seed)
    n, embs = 0, []
    for s, cnt in enumerate(counts):
        e = rng.standard_normal((cnt, E)).astype(np.float32)
        embs.append(e)
        np.save(os.path.join(tmp, "embeddings", f"e{s:03d}.npy"), e)
        pq.write_table(pa.table({"caption": [f"cap {n + i} " + "w " * ((n + i) % 9) for i in range(cnt)]}), os.path.join(tmp, "captions", f"c{s:03d}.parquet"))
        n += cnt
    return np.concatenate(embs)


@pytest.mark.parametrize("world", [1, 2, 3])
def test_piece_pipeline_is_ordered_and_deterministic_across_ranks_and_worker_counts(tmp_path, world):
    from clipcap_amd.train.dataloader import EmbedDataset, pad_tokens
    from clipcap_amd.train.ddp import shard_range
    allemb = _dataset(str(tmp_path))
    B, L = 64, 6
    ref = None
    for par, piece_mb in ((0, 50), (1, 1), (3, 1), (10, 50)):
        per_rank = []
        for r in range(world):
            ds = EmbedDataset(str(tmp_path), batch_size=B, tokenizer=_Tok(), max_token_length=L, rank=r, world_size=world,
                              reader_parallel_pieces=par, reader_max_piece_size=piece_mb)
            got = [(t.clone(), e.clone()) for t, e in ds]
            assert len(got) == len(ds)
            per_rank.append(got)
        ids = []
        for step in range(len(per_rank[0])):
            lo = step * B * world
            hi = min(allemb.shape[0], lo + B * world)
            for r in range(world):
                t, e = per_rank[r][step]
                a, b = shard_range(hi - lo, r, world)
                assert t.dtype == torch.int64 and t.shape == (b - a, L) and e.dtype == torch.float32
                rows = t[:, 0].numpy()
                assert np.array_equal(rows, np.arange(lo + a, lo + b)), (par, r, step)            # this rank's contiguous slice of the global batch
                assert np.array_equal(e.numpy(), allemb[lo + a:lo + b])                          # embeddings travel with their captions
                want = np.stack([pad_tokens(_Tok().encode(f"cap {i} " + "w " * (i % 9)), L) for i in rows[:5]])
                assert np.array_equal(t[:5].numpy(), want)                                       # dataloader.py:41-50: -1 padding / truncation
                ids += rows.tolist()
        if ref is None:
            ref = ids
        assert ids == ref, (par, piece_mb)
    tail = allemb.shape[0] % (B * world)
    assert len(ref) == allemb.shape[0] - (tail if tail < world else 0)


def test_reader_flags_reach_the_dataset(tmp_path):
    """--reader-max-piece-size / --reader-parallel-pieces (clipcap/train/args.py:68-79) are honoured, not just accepted."""
    from clipcap_amd.train.dataloader import get_dataloader
    _dataset(str(tmp_path), counts=(300,))
    ds, E = get_dataloader(str(tmp_path), batch_size=32, tokenizer=_Tok(), reader_max_piece_size=1, reader_parallel_pieces=2, max_token_length=8)
    assert E == 16 and ds.reader_parallel_pieces == 2 and ds.reader_max_piece_size == 1
    # 1 MB of 16-float rows is far more than MAX_BATCHES_PER_PIECE batches: the batch cap bounds the piece
    assert ds.pieces()[0] == (0, 32 * ds.MAX_BATCHES_PER_PIECE)
    ds.reader_max_piece_size = 0
    assert ds.pieces()[0] == (0, 32)                       # never less than one batch


class PicklableTok(_Tok):
    """module-level, so it pickles: the reader then uses worker PROCESSES (spawn), as it does with a real tokenizers / HF tokenizer object."""


def test_process_workers_equal_the_sequential_reader(tmp_path):
    from clipcap_amd.train.dataloader import EmbedDataset
    _dataset(str(tmp_path), counts=(500, 411))
    kw = dict(batch_size=32, max_token_length=6, rank=1, world_size=2, reader_max_piece_size=1)
    seq = [(t.clone(), e.clone()) for t, e in EmbedDataset(str(tmp_path), tokenizer=PicklableTok(), reader_parallel_pieces=0, **kw)]
    ds = EmbedDataset(str(tmp_path), tokenizer=PicklableTok(), reader_parallel_pieces=3, reader_backend="process", **kw)
    try:
        for epoch in range(2):                               # the pool is kept across epochs
            got = list(ds)
            assert len(got) == len(seq) == len(ds)
            for (t0, e0), (t1, e1) in zip(seq, got):
                assert torch.equal(t0, t1) and torch.equal(e0, e1)
        assert ds._pool is not None
    finally:
        ds.close()
    # "auto": this tokenizer pickles -> processes; the local _Tok subclass of another test module would not -> threads
    assert EmbedDataset(str(tmp_path), tokenizer=PicklableTok(), reader_parallel_pieces=2, **kw)._executor(2)[1] is False
