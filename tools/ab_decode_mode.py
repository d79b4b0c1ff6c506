#!/usr/bin/env python
"""A/B of cc_decode_mode bits on ONE box, alternating: beam decode (bench.py's configs[4] workload) with a bit off / on.
usage: ab_decode_mode.py [bit=8] [alternations=3]      (bit 8: fragment-ordered weight image for the K-over-the-waves decode GEMMs)"""
import argparse
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from clipcap_amd import _lib  # noqa: E402

if __name__ == "__main__":
    bit = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    alt = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    l = _lib.lib()
    base = l.cc_decode_mode(-1)
    dev = torch.device("cuda", 0)
    only = os.environ.get("AB_ONLY")           # "on" / "off": one mode, one decode_bench call (for a profiler run)
    for r in range(1 if only else alt):
        for on in ((1,) if only == "on" else (0,) if only == "off" else (0, 1)):
            l.cc_decode_mode((base | bit) if on else (base & ~bit))
            d = bench.decode_bench(argparse.Namespace(batch=0, steps=3, warmup=1, regions=3), dev)
            print(f"bit {bit} {'on ' if on else 'off'}: {d['ms_per_step']:.2f} ms per batch, {d['value']:.0f} tok/s  (regions {d['timed_regions']['ms_per_batch_each']})", flush=True)
    l.cc_decode_mode(base)
