#!/usr/bin/env python
"""Small-grid NT GEMMs (<= 256 tiles: decode M = 320, mapper M = 5120): time + max error vs fp32 matmul.  Run twice with
CC_GEMM_X2=0 / 1 to compare the 4-wave and 8-wave small-grid kernels."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipcap_amd import _lib

lib = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [(320, 4096, 1024), (320, 3072, 1024), (320, 1024, 1024), (320, 1024, 4096), (320, 2304, 768), (320, 3072, 768),
          (5120, 768, 768), (5120, 1536, 768), (5120, 768, 1536), (5120, 7680, 512), (2560, 1024, 1024), (64, 4096, 1024), (128, 4096, 1024)]
for M, N, K in shapes:
    torch.manual_seed(0)
    A = torch.randn(M, K, device="cuda").bfloat16()
    B = torch.randn(N, K, device="cuda").bfloat16()
    Cm = torch.zeros(M, N, device="cuda")
    f = lambda: lib.cc_gemm_op16_f32(0, 0, 0, P(A), K, P(B), K, M, N, K, P(Cm), N, None, 1, st())
    for _ in range(3):
        assert f() == 0
    ref = A.float() @ B.float().t()
    err = (Cm - ref).abs().max().item() / ref.abs().max().item()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"M={M:5d} N={N:5d} K={K:5d}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  relerr {err:.1e}")
