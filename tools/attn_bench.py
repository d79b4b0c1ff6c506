#!/usr/bin/env python
"""Attention kernels alone at the training shapes (GPT-2-small rows: B=256, T=50, 12 heads of 64, causal; mapper rows: B=256, S=20,
8 heads of 96): forward and backward time per call and the HBM-roofline fraction of the bytes each has to move."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipcap_amd import _lib

lib = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(f, iters=200):
    for _ in range(10):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    for (B, S, H, hd, causal) in [(256, 50, 12, 64, 1), (256, 20, 8, 96, 0), (128, 50, 16, 64, 1), (128, 20, 8, 128, 0)]:
        D = H * hd
        qkv = torch.randn(B * S, 3 * D, device="cuda").bfloat16()
        out = torch.empty(B * S, D, dtype=torch.bfloat16, device="cuda")
        lse = torch.empty(B, H, S, device="cuda")
        dout = torch.randn(B * S, D, device="cuda").bfloat16()
        dqkv = torch.empty_like(qkv)
        delta = torch.empty(B * H * S, device="cuda")
        fwd = lambda: lib.cc_attention_fwd(0, P(qkv), B, S, H, hd, causal, P(out), P(lse), st())
        bwd = lambda: lib.cc_attention_bwd(0, P(qkv), P(dout), P(out), P(lse), P(delta), B, S, H, hd, causal, P(dqkv), st())
        assert fwd() == 0 and bwd() == 0
        tf, tb = timeit(fwd), timeit(bwd)
        bytes_f = B * S * D * 2 * 4            # qkv read + out written
        bytes_b = B * S * D * 2 * (3 + 1 + 1 + 3)   # qkv, dO, O read; dqkv written
        print(f"B={B} S={S} H={H} hd={hd} causal={causal}: fwd {tf:.1f} us ({bytes_f / tf / 1e6:.2f} TB/s)  bwd {tb:.1f} us ({bytes_b / tb / 1e6:.2f} TB/s)")


if __name__ == "__main__":
    main()
