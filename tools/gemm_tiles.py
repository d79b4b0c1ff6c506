#!/usr/bin/env python
"""NT GEMM tile-kernel comparison behind the chooser in gemm.hip.h: per shape, the 128x128, 256x192, 256x256 and 320x256 kernels are timed
interleaved (3 rounds, >= 20 ms each, best round kept) so clock drift does not favour one of them."""
import ctypes as C
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipcap_amd import _lib

lib = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [(10240, 50304, 768), (10240, 768, 50304), (12800, 2304, 768), (12800, 768, 2304), (12800, 768, 768), (12800, 3072, 768),
          (12800, 768, 3072), (5120, 768, 768), (5120, 1536, 768), (5120, 768, 1536), (5120, 2304, 768), (6400, 4096, 1024),
          (6400, 1024, 4096), (6400, 3072, 1024), (6400, 1024, 1024), (4096, 4096, 4096), (8192, 8192, 8192)]


def time_mode(mode, A, B, Cm, M, N, K):
    lib.cc_gemm_tile_mode(mode)
    f = lambda: lib.cc_gemm_op16_f32(0, 0, 0, P(A), K, P(B), K, M, N, K, P(Cm), N, None, 1, st())
    assert f() == 0
    iters = max(5, int(20e-3 / (2.0 * M * N * K / 600e12)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    print("| M | N | K | 128x128 us | 256x192 us | 256x256 us | 320x256 us | chooser us | best |")
    print("|---|---|---|---|---|---|---|---|---|")
    for (M, N, K) in SHAPES:
        A = torch.randn(M, K, device="cuda").bfloat16()
        B = torch.randn(N, K, device="cuda").bfloat16()
        Cm = torch.zeros(M, N, device="cuda")
        best = {}
        for _ in range(3):
            for mode in (0, 3, 4, 5, -1):
                t = time_mode(mode, A, B, Cm, M, N, K)
                best[mode] = min(best.get(mode, 1e30), t)
        w = min((0, 3, 4, 5), key=lambda m: best[m])
        fl = 2.0 * M * N * K
        names = {0: '128x128', 3: '256x192', 4: '256x256', 5: '320x256'}
        print(f"| {M} | {N} | {K} | {best[0]:.1f} ({fl / best[0] / 1e6:.0f} TF) | {best[3]:.1f} ({fl / best[3] / 1e6:.0f}) | {best[4]:.1f} ({fl / best[4] / 1e6:.0f}) "
              f"| {best[5]:.1f} ({fl / best[5] / 1e6:.0f}) | {best[-1]:.1f} | {names[w]} |")
    lib.cc_gemm_tile_mode(-1)


if __name__ == "__main__":
    main()
