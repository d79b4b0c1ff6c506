#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table (like `--stats` CSV).

usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--skip-first-us N] > profiles/rNN_*.md
"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
    scols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    q = f"select s.{name_col}, d.end - d.start from {disp} d join {sym} s on d.kernel_id = s.id"
    stats = {}
    for name, dur in cur.execute(q):
        name = re.sub(r"\s+", " ", name)
        a = stats.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in stats.values())
    print(f"| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for name, (n, t, mn, mx) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) < 150 else name[:147] + "..."
        print(f"| `{short}` | {n} | {t / 1e6:.3f} | {t / n / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100.0 * t / total:.2f} |")
    print(f"\ntotal kernel time: {total / 1e6:.3f} ms over {sum(a[0] for a in stats.values())} dispatches")


if __name__ == "__main__":
    main()
