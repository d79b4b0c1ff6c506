#!/usr/bin/env python
"""Randomised parity sweep of the kernels behind the C-ABI test hooks: the assertions of tests/test_gpu_kernels.py re-run on random
shapes (attention forward / backward over sequence lengths 1..100 and head dims 64 / 96 / 128, every NT tile kernel on ragged M / N,
weight-gradient kernels, LayerNorm) for both operand types.  Not part of the test suite (minutes, not seconds):
    python tools/fuzz_kernels.py [seconds] [seed]"""
import os
import random
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import test_gpu_kernels as K


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0, n, fails = time.time(), 0, []
    while time.time() - t0 < budget:
        K.OP[0], K.OP[1] = rng.choice([(0, torch.bfloat16), (1, torch.float16)])
        kind = rng.choice(["attn", "attn", "tile", "tile", "skinny", "layout", "wgrad", "ln"])
        try:
            if kind == "attn":
                hd = rng.choice([64, 96, 128, 64, 96])
                args = (rng.randint(1, 5), rng.randint(1, 100), rng.randint(1, 5), hd, rng.randint(0, 1), 1)
                K.test_attention_fwd_bwd(*args)
            elif kind == "tile":
                mode = rng.choice([3, 4, 5, 6, 7, 7, 7])
                # mode 7 (round 6): the persistent 4-wave kernel — K in multiples of 64 so that both of its rings (K % 192 == 0, K % 128 == 0) and the
                # fall-back to the staggered kernel are drawn; now and then more tiles than CUs (the workgroups then walk several tiles)
                K_ = 64 * rng.randint(1, 40) if mode == 7 else 32 * rng.randint(1, 40)
                big = mode == 7 and rng.random() < 0.15
                args = (mode, 8 * rng.randint(1, 150) * (6 if big else 1), 8 * rng.randint(1, 120) * (4 if big else 1), K_)
                K.test_gemm_nt_256_row_tiles(*args)
            elif kind == "skinny":      # the decode-sized 64-row / 80-row tile kernels (round 5: mode 4 = 80 x 64, K over the waves), single pass and K slices
                ks = rng.choice([1, 1, 1, 2, 3])
                args = (rng.choice([1, 2, 3, 4, 4]), rng.randint(1, 640), 8 * rng.randint(1, 520), 64 * ks * rng.randint(1, 24), ks)
                K.test_gemm_nt_skinny_64_row_tiles(*args)
            elif kind == "layout":
                al, bl = rng.choice([(0, 0), (0, 1), (1, 1)])
                args = (al, bl, 8 * rng.randint(1, 130), 8 * rng.randint(1, 100), 8 * rng.randint(1, 140))
                K.test_gemm_layouts(*args)
            elif kind == "wgrad":
                args = (rng.choice([4, 0, -1]), rng.randint(1, 3000), 8 * rng.randint(1, 100), 8 * rng.randint(1, 100))
                K.test_gemm_wgrad_kernels(*args)
            else:
                args = (rng.randint(1, 300), 4 * rng.randint(4, 400))
                K.test_layernorm_fwd(*args)
        except BaseException as e:      # pytest.skip raises too: count only assertion / runtime failures
            if type(e).__name__ in ("Skipped",):
                continue
            fails.append((kind, K.OP[0], args, repr(e)[:200]))
            print("FAIL", kind, "op", K.OP[0], args, repr(e)[:200], flush=True)
        n += 1
    print(f"{n} cases in {time.time() - t0:.0f} s, {len(fails)} failures")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
