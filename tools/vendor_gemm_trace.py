#!/usr/bin/env python
"""Per-shape GPU-side comparison with the vendor GEMM: runs tools/vendor_gemm_yardstick.py ONE SHAPE PER PROCESS under
`rocprofv3 --kernel-trace` and reads, from the trace, which hipBLASLt kernel (macro tile, stream-K or not) the vendor library picks for the
shape and what both kernels take ON THE GPU (median dispatch duration) — the yardstick's own event timings include the host's launch cost,
which for the 10-20 us shapes is most of the vendor figure.  A measuring stick only: nothing in clipcap_amd calls a vendor GEMM.

usage (GPU box, from the repo root): python tools/vendor_gemm_trace.py > gpurun_out/rNN_vendor_gemm_trace.md"""
import os
import re
import shutil
import sqlite3
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dispatches(dbpath):
    db = sqlite3.connect(dbpath)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    scols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    out = {}
    for name, dur in cur.execute(f"select s.{name_col}, d.end - d.start from {disp} d join {sym} s on d.kernel_id = s.id"):
        out.setdefault(re.sub(r"\s+", " ", name), []).append(dur / 1e3)
    return out


def short_vendor(name):
    mt = re.search(r"MT(\d+x\d+x\d+)", name)
    flags = [f for f in ("Custom", "SK3", "LDSB0", "LDSB1", "DTVA1", "DTVB1") if f in name]
    wg = re.search(r"_WG(\d+_\d+_\d+)", name)
    return "MT" + (mt.group(1) if mt else "?") + (" " + " ".join(flags) if flags else "") + (" WG" + wg.group(1) if wg else "")


def short_ours(name):
    m = re.search(r"\d+(gemm_\w+?_kernel)I(.*?)EEvPK", name)
    if not m:
        return name[:60]
    args = re.sub(r"NS_\d+(Epi\w+?)E(?=L|$)", r"\1,", m.group(2))
    args = re.sub(r"Li(\d+)E", r"\1,", args).replace("Lb0E", "0,").replace("Lb1E", "1,")
    return f"{m.group(1)}<{args.rstrip(',')}>"


def main():
    from tools.vendor_gemm_yardstick import SHAPES
    tmp = "/tmp/vgt"
    print("| site | M | N | K | vendor kernel (hipBLASLt's pick) | vendor GPU us (TFLOP/s) | ours (fp32-out hook) kernel | ours GPU us (TFLOP/s) | ours / vendor |")
    print("|---|---|---|---|---|---|---|---|---|")
    env = dict(os.environ, TMPDIR="/tmp")
    for (label, M, N, K) in SHAPES:
        shutil.rmtree(tmp, ignore_errors=True)
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "rocpd", "-d", tmp, "--", sys.executable,
                            os.path.join(ROOT, "tools", "vendor_gemm_yardstick.py"), "--exact", label], cwd="/tmp", env=env, capture_output=True, text=True)
        dbs = [os.path.join(d, f) for d, _, fs in os.walk(tmp) for f in fs if f.endswith(".db")]
        if r.returncode != 0 or not dbs:
            print(f"| {label} | {M} | {N} | {K} | trace failed (rc {r.returncode}) | | | | |", flush=True)
            continue
        d = dispatches(dbs[0])
        # the bf16 vendor GEMM is the most-called Cijk kernel (the fp32 check GEMM runs once); ours the most-called cc gemm kernel
        ven = max(((n, v) for n, v in d.items() if "Cijk" in n), key=lambda kv: len(kv[1]), default=None)
        our = max(((n, v) for n, v in d.items() if "gemm_" in n and "cc_" in n), key=lambda kv: len(kv[1]), default=None)
        fl = 2.0 * M * N * K
        if not ven or not our:
            print(f"| {label} | {M} | {N} | {K} | kernels not found | | | | |", flush=True)
            continue
        tv, to = statistics.median(ven[1]), statistics.median(our[1])
        print(f"| {label} | {M} | {N} | {K} | {short_vendor(ven[0])} | {tv:.1f} ({fl / tv / 1e6:.0f}) | {short_ours(our[0])} | {to:.1f} ({fl / to / 1e6:.0f}) | {to / tv:.2f} |", flush=True)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
