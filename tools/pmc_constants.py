#!/usr/bin/env python
"""Turns the four rocprofv3 --pmc passes of tools/profile_round.sh (FETCH_SIZE / WRITE_SIZE on the training step and on the beam decode,
one counter per pass) into profiles/pmc_constants.json — the offline traffic figures bench.py quotes — stamped with the hash of the kernel
sources they were measured on (bench.kernel_source_hash): bench.py reports them as null once the sources change.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> x2; both
counters are in KB.

usage: pmc_constants.py TAG train_fetch.db train_write.db TRAIN_STEPS decode_fetch.db decode_write.db DECODES POSITIONS
"""
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
    q = (f"select s.{name_col}, e.value from {pe} e join {ip} p on e.pmc_id = p.id "
         f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id")
    agg = {}
    for kname, val in cur.execute(q):
        a = agg.setdefault(kname, [0, 0.0])
        a[0] += 1
        a[1] += val
    return agg


def main():
    tag, tf, tw, steps, df, dw, decodes, positions = sys.argv[1:9]
    steps, decodes, positions = int(steps), int(decodes), int(positions)
    from bench import kernel_source_hash
    h = kernel_source_hash()
    kb = 1024.0
    f, w = per_kernel(tf), per_kernel(tw)
    gem_f = sum(v[1] for k, v in f.items() if "gemm_" in k) * 2 * kb / steps
    gem_w = sum(v[1] for k, v in w.items() if "gemm_" in k) * kb / steps
    lm = [k for k in f if "EpiLMHead" in k]
    lm_b = sum(f[k][1] / f[k][0] * 2 * kb + w[k][1] / w[k][0] * kb for k in lm) if lm else None
    dfk, dwk = per_kernel(df), per_kernel(dw)
    dec = (sum(v[1] for v in dfk.values()) * 2 + sum(v[1] for v in dwk.values())) * kb / (decodes * positions)
    att = [k for k in dfk if "k_decode_attn" in k]
    att_b = sum(dfk[k][1] for k in att) * 2 * kb / max(1, sum(dfk[k][0] for k in att))
    out = {
        "train_gemm_bytes_per_step": {"bytes": int(gem_f + gem_w), "fetch_bytes": int(gem_f), "write_bytes": int(gem_w), "kernel_source_hash": h,
                                      "source": f"profiles/{tag}_e_pmc_fetch_write_train.md (offline PMC: sum over the gemm_* kernels of one step)"},
        "lmhead_fwd_bytes_per_launch": {"bytes": int(lm_b) if lm_b else None, "kernel_source_hash": h,
                                        "source": f"profiles/{tag}_e_pmc_fetch_write_train.md (offline PMC, lm_head forward launch)"},
        "decode_bytes_per_position": {"bytes": int(dec), "kernel_source_hash": h, "attention_fetch_bytes_per_launch": int(att_b),
                                      "source": f"profiles/{tag}_f_pmc_fetch_write_decode.md (offline PMC: every kernel of {decodes} decodes / {decodes * positions} positions)"},
    }
    with open(os.path.join(ROOT, "profiles", "pmc_constants.json"), "w") as fjson:
        json.dump(out, fjson, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
