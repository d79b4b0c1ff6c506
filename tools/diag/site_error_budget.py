#!/usr/bin/env python
"""Which GEMM sites put the 16-bit operand modes outside the 1e-3 logits bar?  

CPU tool on the oracle (oracle/clipcap_oracle.py, per-site rounding modes) and the reference's own fp32 logits in the committed
full-depth fixtures (tests/golden/config{2,3,4}_full.npz: rows / columns sampled by the generator).  For a base operand type
(fp16 or bf16) it prints, per fixture, max |logit - reference fp32| over the loss-relevant rows for

  A. ONE site class at the base type, every other site exact        -> what each site costs on its own
  B. every site at the base type, ONE site class promoted to split operands (hi + lo of the same type) -> what promoting it buys
  C. greedy: promote the site that helps most, repeat until the bar (1e-3) is met

Site classes (oracle._m): mapper (all its GEMMs and its attention), c_attn, attn (stored qkv + probabilities of P V), attn.c_proj,
c_fc, mlp.c_proj, lm_head.  usage: python tools/diag/site_error_budget.py [--base fp16|bf16] [--fixtures config2_full ...]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import clipcap_oracle as O  # noqa: E402
from tests.seeded import sample_idx  # noqa: E402
from tests.util import load_golden, seeded_full_model  # noqa: E402

SITES = ["mapper", "c_attn", "attn", "attn.c_proj", "c_fc", "mlp.c_proj", "lm_head"]


def case(name):
    g = load_golden(name)
    sd, cfg, dims = seeded_full_model(g)
    tokens, embeds = torch.from_numpy(g["in.tokens"]), torch.from_numpy(g["in.embeds"])
    L, V, cap, B = dims["L"], dims["V"], tokens.shape[1], tokens.shape[0]
    valid = torch.cat((torch.ones(B, L, dtype=torch.bool), tokens.ge(0)), dim=1)
    cols = sample_idx(V, 1024)
    ref = torch.from_numpy(g["logits.cols"])

    def err(rb):
        with torch.no_grad():
            lg = O.clipcap_logits(sd, tokens.clamp_min(0), embeds, cfg=cfg, rb=rb)
        return float(((lg[:, :, cols] - ref) * valid[:, :, None]).abs().max())
    return err, float(g["logits.absmax"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--fixtures", nargs="*", default=["config2_full", "config3_full", "config4_full"])
    ap.add_argument("--bar", type=float, default=1e-3)
    a = ap.parse_args()
    base = a.base
    split = "fp16x2" if base == "fp16" else "bf16x3"
    torch.set_num_threads(max(1, (os.cpu_count() or 8)))
    for name in a.fixtures:
        err, amax = case(name)
        e_exact, e_all, e_split = err(False), err(base), err(split)
        print(f"\n## {name}: |logits| max {amax:.2f}; oracle fp32 vs reference {e_exact:.2e}; every site {base} {e_all:.2e}; every site split ({split}) {e_split:.2e}; bar {a.bar:.0e}")
        print(f"| site class | A: only this site {base} | B: all {base}, this site split |")
        print("|---|---|---|")
        for s in SITES:
            ea = err({"default": False, s: base})
            eb = err({"default": base, s: split})
            print(f"| {s} | {ea:.2e} | {eb:.2e} |", flush=True)
        promoted, cur = [], e_all
        while cur > a.bar and len(promoted) < len(SITES):
            best = None
            for s in SITES:
                if s in promoted:
                    continue
                mode = {"default": base, **{q: split for q in promoted + [s]}}
                e = err(mode)
                if best is None or e < best[1]:
                    best = (s, e)
            promoted.append(best[0]); cur = best[1]
            print(f"C: promoted {promoted} -> {cur:.2e}", flush=True)


if __name__ == "__main__":
    main()
