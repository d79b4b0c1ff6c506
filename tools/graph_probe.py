#!/usr/bin/env python
"""Does replaying a captured hipGraph shorten a decode step?  One cc_decode_fwd (170 dependent kernels, 320 rows, GPT-2-medium, position
40) eager vs captured with torch.cuda.CUDAGraph."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipcap_amd.engine import DecodeSession
from clipcap_amd.model.gpt2 import GPT2LM

dev = torch.device("cuda", 0)
torch.manual_seed(0)
lm = GPT2LM(n_embd=1024, n_layer=24, n_head=16, vocab_size=50257, n_positions=1024).to(dev)
R = 320
sess = DecodeSession(lm.engine, R, 77)
sess.forward(torch.randn(R, 40, 1024, device=dev) * 0.3)
x = torch.randn(R, 1, 1024, device=dev) * 0.3
pos = sess.pos


def step():
    sess.pos = pos
    return sess.forward(x)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
torch.cuda.synchronize()
print(f"eager   : {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per decode step")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = step()
torch.cuda.synchronize()
ref = step().clone()
g.replay()
torch.cuda.synchronize()
print("graph output equals eager:", bool(torch.equal(out, ref)))
t0 = time.perf_counter()
for _ in range(50):
    g.replay()
torch.cuda.synchronize()
print(f"graph   : {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per decode step")
