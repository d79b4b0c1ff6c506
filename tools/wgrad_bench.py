#!/usr/bin/env python
"""Weight-gradient GEMM (dW += X^T Y, both operands K-strided) through cc_gemm_wgrad: the 256x256 DMA + transpose-read kernel
(tile mode 4 = forced) against the register-staged 128x128 kernel (tile mode 0) and the chooser (-1: 256x256 from ~30 GFLOP, else the
128x128 DMA + transpose-read kernel), interleaved, best of 3."""
import ctypes as C
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipcap_amd import _lib

lib = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [(5120, 768, 2304), (5120, 768, 768), (5120, 768, 1536), (5120, 1536, 768), (256, 512, 7680),
          (12800, 768, 2304), (12800, 768, 768), (12800, 768, 3072), (12800, 3072, 768), (10240, 50304, 768),
          (6400, 1024, 3072), (6400, 1024, 1024), (6400, 1024, 4096), (6400, 4096, 1024)]


def main():
    scratch = torch.empty(lib.cc_wgrad_scratch_bytes(), dtype=torch.uint8, device="cuda")
    print("| K | Mw | Nw | 128x128 register-staged us | 256x256 DMA+tr us | chooser us |")
    print("|---|---|---|---|---|---|")
    for (K, Mw, Nw) in SHAPES:
        X = torch.randn(K, Mw, device="cuda").bfloat16()
        Y = torch.randn(K, Nw, device="cuda").bfloat16()
        dW = torch.zeros(Mw, Nw, device="cuda")
        best = {}
        fl = 2.0 * K * Mw * Nw
        iters = max(5, int(10e-3 / (fl / 500e12)))
        for _ in range(3):
            for mode in (0, 4, -1):
                lib.cc_gemm_tile_mode(mode)
                f = lambda: lib.cc_gemm_wgrad(0, P(X), Mw, P(Y), Nw, Mw, Nw, K, P(dW), Nw, P(scratch), st())
                assert f() == 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    f()
                e1.record()
                torch.cuda.synchronize()
                best[mode] = min(best.get(mode, 1e30), e0.elapsed_time(e1) / iters * 1e3)
        print(f"| {K} | {Mw} | {Nw} | {best[0]:.1f} ({fl / best[0] / 1e6:.0f} TF) | {best[4]:.1f} ({fl / best[4] / 1e6:.0f} TF) | {best[-1]:.1f} |")
    lib.cc_gemm_tile_mode(-1)


if __name__ == "__main__":
    main()
