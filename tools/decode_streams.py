#!/usr/bin/env python
"""Experiment: beam decode of 64 prefixes as ONE batch vs as 2 / 4 sub-batches decoded concurrently on separate HIP streams (one host
thread per stream).  Sub-batches give smaller kernels (fewer workgroups each), whose launch floors and tails can overlap."""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from clipcap_amd.inference.base import generate_beam_tokens
from clipcap_amd.model.gpt2 import GPT2LM

dev = torch.device("cuda", 0)
torch.manual_seed(1234)
lm = GPT2LM(n_embd=1024, n_layer=24, n_head=16, vocab_size=50257, n_positions=1024).to(dev)
model = SimpleNamespace(language_model=lm)
S = 64
prefix = torch.randn(S, 10, 1024, device=dev) * 0.5
generate_beam_tokens(model, prefix, 5, 67, 1.0, 50256)
torch.cuda.synchronize()


def run(parts, reps=5):
    chunks = prefix.chunk(parts)
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    outs = [None] * parts

    def work(i):
        with torch.cuda.stream(streams[i]):
            for _ in range(reps):
                outs[i] = generate_beam_tokens(model, chunks[i], 5, 67, 1.0, 50256)
    for i in range(parts):              # warm (workspace allocations per row count)
        with torch.cuda.stream(streams[i]):
            generate_beam_tokens(model, chunks[i], 5, 67, 1.0, 50256)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(parts)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{parts} stream(s) x {S // parts} prefixes: {dt * 1e3:7.2f} ms per 64 prefixes")
    return outs


a = run(1)
b = run(2)
c = run(4)
ta = a[0][0]
tb = torch.cat([o[0] for o in b])
print("tokens equal (1 vs 2 streams):", bool(torch.equal(ta, tb)))
seq = [generate_beam_tokens(model, ch, 5, 67, 1.0, 50256) for ch in prefix.chunk(2)]
torch.cuda.synchronize()
print("2 sub-batches sequential == concurrent:", all(bool(torch.equal(s[0], o[0])) and bool(torch.equal(s[1], o[1])) for s, o in zip(seq, b)))
