#!/usr/bin/env python
"""Probe: can the mapper backward's weight-gradient work hide under its input-gradient chain on a second HIP stream?  The chain is ~16 us
single-round launches that leave the MFMA pipe ~80 % idle; the weight gradients are throughput-bound launches off the dependency chain.  Two
kernels share a CU only if their LDS fits together (160 KiB): the product's small-grid GEMM (128 KiB) and grouped weight-gradient kernel
(128 KiB) cannot, the 2-stage 128 x 128 kernels (64 KiB each, cc_gemm_tile_mode 0) can.  Measures, per tile mode: chain alone, weight
gradients alone, both streams together.  together ~ max(alone) = overlap; together ~ sum = none.

usage (GPU box): python tools/stream_overlap_probe.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipcap_amd import _lib

lib = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
dev = "cuda"
M, D, Hm = 5120, 768, 1536
torch.manual_seed(0)
bf = lambda *s: torch.randn(*s, device=dev).bfloat16()
# one mapper layer's backward GEMM shapes: input-gradient chain (NT) and weight gradients (TT)
chain = [(bf(M, D), bf(Hm, D), M, Hm, D), (bf(M, Hm), bf(D, Hm), M, D, Hm), (bf(M, D), bf(D, D), M, D, D), (bf(M, 3 * D), bf(D, 3 * D), M, D, 3 * D)]
couts = [torch.zeros(m, n, device=dev) for (_, _, m, n, _) in chain]
wg = [(bf(M, D), bf(M, Hm)), (bf(M, Hm), bf(M, D)), (bf(M, D), bf(M, D)), (bf(M, 3 * D), bf(M, D))]
wouts = [torch.zeros(x.shape[1], y.shape[1], device=dev) for (x, y) in wg]
scratch = torch.empty(lib.cc_wgrad_scratch_bytes(), dtype=torch.uint8, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
LAYERS = 8


def run_chain(st):
    for _ in range(LAYERS):
        for (a, b, m, n, k), c in zip(chain, couts):
            assert lib.cc_gemm_op16_f32(0, 0, 0, P(a), k, P(b), k, m, n, k, P(c), n, None, 1, C.c_void_p(st.cuda_stream)) == 0


def run_wgrad(st):
    for _ in range(LAYERS):
        for (x, y), w in zip(wg, wouts):
            assert lib.cc_gemm_wgrad(0, P(x), x.shape[1], P(y), y.shape[1], x.shape[1], y.shape[1], M, P(w), y.shape[1], P(scratch), C.c_void_p(st.cuda_stream)) == 0


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e3


def both():
    run_chain(sa)
    run_wgrad(sb)


print("| tile mode | chain alone us (8 layers x 4 NT GEMMs) | weight gradients alone us (8 x 4 TT GEMMs) | both streams us | sum | max |")
print("|---|---|---|---|---|---|")
for mode, name in ((-1, "product chooser (128-KiB kernels)"), (0, "2-stage 128 x 128 kernels only (64 KiB each)")):
    old = lib.cc_gemm_tile_mode(mode)
    try:
        both()
        torch.cuda.synchronize()
        ta = timed(lambda: run_chain(sa))
        tb = timed(lambda: run_wgrad(sb))
        tc = timed(both)
    finally:
        lib.cc_gemm_tile_mode(old)
    print(f"| {name} | {ta:.0f} | {tb:.0f} | {tc:.0f} | {ta + tb:.0f} | {max(ta, tb):.0f} |", flush=True)
