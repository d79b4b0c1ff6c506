#!/usr/bin/env python
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database.  usage: rocpd_pmc.py db [name-substring]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda p: next(x for x in tabs if x.startswith(p))
    pe, ip, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    kcols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    q = (f"select s.{name_col}, p.name, e.value from {pe} e join {ip} p on e.pmc_id = p.id "
         f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id")
    agg = {}
    for kname, cname, val in cur.execute(q):
        if pat and pat not in kname:
            continue
        a = agg.setdefault((kname, cname), [0, 0.0])
        a[0] += 1
        a[1] += val
    for (kname, cname), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{cname:12s} avg/dispatch {tot / n:14.1f}  dispatches {n:5d}  {kname[:110]}")


if __name__ == "__main__":
    main()
