#!/usr/bin/env python
"""Vendor-GEMM yardstick: `torch.matmul` (hipBLASLt / rocBLAS through PyTorch) on the NT shapes of the model, timed
beside this library's own kernels.  A MEASURING STICK ONLY: nothing in clipcap_amd links, loads or calls a vendor GEMM; this script
is the one place the two meet, so that "how fast can a K = 768 launch be on this chip" has an answer that is not the builder's own.

Output: a markdown table (stdout) -> profiles/rNN_*_vendor_gemm_yardstick.md.
  * vendor bf16: C[M,N] bf16 = A[M,K] bf16 . B[N,K]^T (the layout every forward / dgrad GEMM of the library sees: both K-contiguous)
  * ours f32-out: cc_gemm_op16_f32 (chooser's kernel, fp32 C) — the only bare GEMM hook of the C ABI; its epilogue stores 2x the bytes
  * ours bf16-out where an engine call site exists is in the kernel traces (profiles/r05_*_kernel_stats.md)
Each pair is timed interleaved (3 rounds, >= 15 ms per leg, best round kept)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipcap_amd import _lib

lib = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

# (label, M, N, K)
SHAPES = [
    ("lm_head fwd", 10240, 50304, 768), ("lm_head dgrad", 10240, 768, 50304),
    ("gpt2 c_attn", 12800, 2304, 768), ("gpt2 c_attn dgrad", 12800, 768, 2304), ("gpt2 attn.c_proj", 12800, 768, 768),
    ("gpt2 c_fc / gelu' dgrad", 12800, 3072, 768), ("gpt2 mlp.c_proj / c_fc dgrad", 12800, 768, 3072),
    ("mapper project", 5120, 768, 768), ("mapper fc1", 5120, 1536, 768), ("mapper fc2", 5120, 768, 1536), ("mapper qkv", 5120, 2304, 768),
    ("mapper qkv dgrad", 5120, 768, 2304),
    ("medium c_fc", 6400, 4096, 1024), ("medium mlp.c_proj", 6400, 1024, 4096), ("medium c_attn", 6400, 3072, 1024),
    ("medium attn.c_proj", 6400, 1024, 1024),
    ("decode c_attn", 320, 3072, 1024), ("decode attn.c_proj", 320, 1024, 1024), ("decode c_fc", 320, 4096, 1024),
    ("decode mlp.c_proj", 320, 1024, 4096), ("decode lm_head", 320, 50304, 1024),
    ("square 4096", 4096, 4096, 4096), ("square 8192", 8192, 8192, 8192),
]


def timed(f, flops):
    f()
    iters = max(5, int(15e-3 / (flops / 500e12)))
    iters = min(iters, 2000)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    only = [a for a in sys.argv[1:] if a != "--exact"]
    exact = "--exact" in sys.argv[1:]            # tools/vendor_gemm_trace.py: one shape per process, labels matched whole
    print("| site | M | N | K | vendor bf16-out us (TFLOP/s) | vendor fp32-acc check | ours (chooser, fp32 out) us (TFLOP/s) | ours / vendor |")
    print("|---|---|---|---|---|---|---|---|")
    lib.cc_gemm_tile_mode(-1)
    for (label, M, N, K) in SHAPES:
        if only and not (label in only if exact else any(o in label for o in only)):
            continue
        torch.manual_seed(0)
        A = torch.randn(M, K, device="cuda").bfloat16()
        B = torch.randn(N, K, device="cuda").bfloat16()
        Cv = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        Cm = torch.zeros(M, N, device="cuda")
        Bt = B.t()
        fv = lambda: torch.matmul(A, Bt, out=Cv)
        fo = lambda: lib.cc_gemm_op16_f32(0, 0, 0, P(A), K, P(B), K, M, N, K, P(Cm), N, None, 1, st())
        assert fo() == 0
        fv()
        fl = 2.0 * M * N * K
        rows = min(M, 256)
        ref = A[:rows].float() @ B.float().t()
        ev = (Cv[:rows].float() - ref).abs().max().item() / ref.abs().max().item()
        eo = (Cm[:rows] - ref).abs().max().item() / ref.abs().max().item()
        best = {}
        for _ in range(3):
            for k, f in (("v", fv), ("o", fo)):
                best[k] = min(best.get(k, 1e30), timed(f, fl))
        print(f"| {label} | {M} | {N} | {K} | {best['v']:.1f} ({fl / best['v'] / 1e6:.0f}) | vendor {ev:.1e} / ours {eo:.1e} "
              f"| {best['o']:.1f} ({fl / best['o'] / 1e6:.0f}) | {best['o'] / best['v']:.2f} |", flush=True)


if __name__ == "__main__":
    main()
