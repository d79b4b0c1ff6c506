#!/usr/bin/env python
"""Micro-benchmark of the bf16 MFMA GEMM through the C ABI test hook (fp32-out epilogue): TFLOP/s per shape/layout."""
import ctypes as C
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipcap_amd import _lib

lib = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(al, bl, M, N, K, ks, iters=20):
    A = torch.randn((K, M) if al else (M, K), device="cuda").bfloat16()
    B = torch.randn((K, N) if bl else (N, K), device="cuda").bfloat16()
    Cm = torch.zeros(M, N, device="cuda")
    f = lambda: lib.cc_gemm_op16_f32(0, al, bl, P(A), A.shape[1], P(B), B.shape[1], M, N, K, P(Cm), N, None, ks, st())
    for _ in range(3):
        assert f() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"al={al} bl={bl} M={M:6d} N={N:6d} K={K:6d} ksplit={ks:2d}: {ms * 1e3:9.1f} us  {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s")


if __name__ == "__main__":
    shapes = [
        (0, 0, 5120, 768, 768, 1), (0, 0, 5120, 2304, 768, 1), (0, 0, 5120, 1536, 768, 1), (0, 0, 5120, 768, 1536, 1),
        (0, 1, 5120, 768, 768, 1), (0, 1, 12800, 2304, 768, 1), (0, 1, 12800, 3072, 768, 1), (0, 1, 12800, 768, 3072, 1),
        (0, 0, 12800, 3072, 768, 1), (0, 0, 10240, 50304, 768, 1), (0, 1, 10240, 768, 50304, 1),
        (1, 1, 768, 1536, 5120, 1), (1, 1, 768, 1536, 5120, 3), (1, 1, 768, 1536, 5120, 7), (1, 1, 768, 768, 5120, 14),
        (1, 1, 2304, 768, 5120, 4), (1, 1, 768, 768, 5120, 1), (0, 0, 4096, 4096, 4096, 1), (0, 1, 4096, 4096, 4096, 1), (1, 1, 4096, 4096, 4096, 1),
    ]
    for s in shapes:
        run(*s)
