#!/usr/bin/env python
"""MFMA utilisation per kernel from a rocprofv3 rocpd database collected with `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` (its own pass,
with --kernel-trace only; tools/profile_round.sh).  Method (profiles/r01_o, r02_o, r03_h): SQ counters come per shader-engine instance
(32 SIMDs each), GRBM_GUI_ACTIVE per XCD, so for one dispatch

    MFMA pipe busy = (mean over instances of SQ_VALU_MFMA_BUSY_CYCLES / 32) / (mean over instances of GRBM_GUI_ACTIVE)

i.e. the fraction of the dispatch's active cycles in which a SIMD's matrix pipe was executing, averaged over the chip.

usage: mfma_util.py LABEL=db [LABEL=db ...] [--json profiles/pmc_constants.json] [--min-cycles N]
Prints one markdown table per label; with --json, stores {label: {kernel symbol: busy}} + the time-weighted GEMM-family figure under
"mfma_busy" in that file, stamped with bench.kernel_source_hash() like the traffic constants."""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_kernel(dbpath):
    db = sqlite3.connect(dbpath)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda p: next(x for x in tabs if x.startswith(p))  # noqa: E731
    pe, ip, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    scols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    q = (f"select d.event_id, s.{name_col}, p.name, e.value from {pe} e join {ip} p on e.pmc_id = p.id "
         f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id")
    disp = {}
    for ev, kname, cname, val in cur.execute(q):
        d = disp.setdefault(ev, [kname, [0, 0.0], [0, 0.0]])
        slot = d[1] if "MFMA_BUSY" in cname else d[2] if "GUI_ACTIVE" in cname else None
        if slot is not None:
            slot[0] += 1
            slot[1] += val
    agg = {}
    for kname, (n_sq, sq), (n_g, g) in disp.values():
        if not n_sq or not n_g or g <= 0:
            continue
        active = g / n_g
        busy = (sq / n_sq / 32.0) / active
        a = agg.setdefault(kname, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += active
        a[2] += busy * active          # cycle-weighted
    return {k: {"launches": n, "active_cycles_per_launch": act / n, "mfma_busy": wb / act} for k, (n, act, wb) in agg.items()}


def main():
    argv = sys.argv[1:]
    jpath, min_cycles = None, 20000.0
    if "--json" in argv:
        i = argv.index("--json")
        jpath = argv[i + 1]
        del argv[i:i + 2]
    if "--min-cycles" in argv:
        i = argv.index("--min-cycles")
        min_cycles = float(argv[i + 1])
        del argv[i:i + 2]
    out = {}
    for spec in argv:
        label, db = spec.split("=", 1)
        k = per_kernel(db)
        rows = sorted(k.items(), key=lambda kv: -kv[1]["active_cycles_per_launch"] * kv[1]["launches"])
        print(f"\n## {label}\n\n| kernel | launches | GUI_ACTIVE cycles / launch | MFMA pipe busy |\n|---|---|---|---|")
        for name, v in rows:
            if v["mfma_busy"] < 0.005 and v["active_cycles_per_launch"] < min_cycles:
                continue
            print(f"| `{name[:150]}` | {v['launches']} | {v['active_cycles_per_launch']:,.0f} | {100 * v['mfma_busy']:.1f} % |")
        gem = {n: v for n, v in k.items() if "gemm_" in n}
        tot = sum(v["active_cycles_per_launch"] * v["launches"] for v in gem.values())
        fam = sum(v["mfma_busy"] * v["active_cycles_per_launch"] * v["launches"] for v in gem.values()) / tot if tot else None
        allc = sum(v["active_cycles_per_launch"] * v["launches"] for v in k.values())
        whole = sum(v["mfma_busy"] * v["active_cycles_per_launch"] * v["launches"] for v in k.values()) / allc if allc else None
        if fam is not None:
            print(f"\nGEMM family (cycle-weighted over its {sum(v['launches'] for v in gem.values())} launches): **{100 * fam:.1f} %**; every kernel of the run: {100 * whole:.1f} %")
        out[label] = {"gemm_family": fam, "all_kernels": whole,
                      "per_kernel": {n: round(v["mfma_busy"], 4) for n, v in rows if v["mfma_busy"] >= 0.005}}
    if jpath:
        from bench import kernel_source_hash
        try:
            with open(jpath) as f:
                allj = json.load(f)
        except (OSError, ValueError):
            allj = {}
        allj["mfma_busy"] = dict(out, kernel_source_hash=kernel_source_hash(),
                                 source="SQ_VALU_MFMA_BUSY_CYCLES / 32 / GRBM_GUI_ACTIVE per dispatch, cycle-weighted (tools/mfma_util.py; profiles/*_mfma_utilisation.md)")
        with open(jpath, "w") as f:
            json.dump(allj, f, indent=1)


if __name__ == "__main__":
    main()
