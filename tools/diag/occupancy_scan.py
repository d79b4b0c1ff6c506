#!/usr/bin/env python
"""Residency / wave-quantisation scan of a rocprofv3 kernel trace (csv): for every kernel symbol, workgroups per launch, waves per workgroup, the
workgroups one CU can hold (LDS 160 KiB, 512 registers per SIMD lane, 8 waves per SIMD slots) and the launch's rounds = workgroups / (256 CUs x resident)
— a fractional part just above .0 means an almost empty last round, `resident` = 1 with small workgroups means one wave per SIMD.
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d out -- python bench.py --steps 2 --warmup 1 --regions 1 --no-cpu-baseline --no-sub-benches --no-roofline-pass
    python tools/diag/occupancy_scan.py out/**/*kernel_trace.csv"""
import csv
import sys
from collections import defaultdict


def main():
    agg = defaultdict(lambda: [0, 0.0, None])
    for path in sys.argv[1:]:
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"]
            wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
            grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
            lds = int(r.get("LDS_Block_Size", 0) or 0)
            vg = int(r.get("VGPR_Count", 0) or 0) + int(r.get("Accum_VGPR_Count", 0) or 0)
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            a = agg[(name, grid // wg, wg, lds, vg)]
            a[0] += 1
            a[1] += dur
    rows = []
    for (name, nwg, wg, lds, vg), (n, us, _) in agg.items():
        waves = (wg + 63) // 64
        by_lds = (160 * 1024) // lds if lds else 99
        per_simd = max(1, 512 // max(vg, 1)) if vg else 8
        by_reg = max(1, (min(per_simd, 8) * 4) // waves) if waves <= 4 * min(per_simd, 8) else 0
        res = max(1, min(by_lds, by_reg, 32 // waves if waves <= 32 else 1))
        rounds = nwg / (256.0 * res)
        rows.append((us, n, us / n, nwg, waves, lds, vg, res, rounds, name))
    rows.sort(reverse=True)
    print("| total us | calls | avg us | workgroups | waves/wg | LDS B | regs | resident wg/CU | rounds | kernel |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for us, n, avg, nwg, waves, lds, vg, res, rounds, name in rows[:40]:
        print(f"| {us:.0f} | {n} | {avg:.1f} | {nwg} | {waves} | {lds} | {vg} | {res} | {rounds:.2f} | `{name[:90]}` |")


if __name__ == "__main__":
    main()
