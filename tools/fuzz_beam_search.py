#!/usr/bin/env python
"""Randomised end-to-end beam search: random small GPT-2s (width 32..128, 1-2 layers, vocabulary 50..400), 1-4 prefixes, beam 1..8,
temperature, stop token, entry length; the product's KV-cached batched generate_beam_tokens against the oracle's per-sample full
re-forward search with the kernels' rounding points.  Beam search is chaotic (one near-tie changes which captions survive), so a
mismatch only counts when the oracle's own search is stable: the rounding-point oracle and the exact-arithmetic oracle must agree
on the caption, otherwise the case is noise by construction.  Not part of the test suite:
    python tools/fuzz_beam_search.py [seconds] [seed]"""
import os
import random
import sys
import time
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipcap_amd.inference.base import generate_beam_tokens
from clipcap_amd.model.gpt2 import GPT2LM
from oracle import clipcap_oracle as O


def one(rng):
    hd = rng.choice([16, 32, 64])
    n_head = rng.choice([1, 2, 4])
    D, NL = hd * n_head, rng.randint(1, 2)
    V = rng.randint(50, 400)
    S, L0, beam, entry = rng.randint(1, 4), rng.randint(1, 6), rng.randint(1, 8), rng.randint(3, 12)
    temp, stop = round(rng.uniform(0.7, 1.3), 2), rng.randrange(V)
    args = dict(D=D, n_head=n_head, NL=NL, V=V, S=S, L0=L0, beam=beam, entry=entry, temp=temp, stop=stop)
    torch.manual_seed(rng.randrange(1 << 30))
    lm = GPT2LM(n_embd=D, n_layer=NL, n_head=n_head, vocab_size=V, n_positions=32).to("cuda")
    with torch.no_grad():
        for n_, p_ in lm.named_parameters():            # spread the logits: random-init GPT-2s are nearly uniform, i.e. all ties
            if "wte" in n_:
                p_.mul_(8.0)
    sd = {"language_model." + k: v.detach().cpu().float() for k, v in lm.state_dict().items() if "lm_head" not in k}
    pref = torch.randn(S, L0, D) * 0.7
    toks, scores, lens = generate_beam_tokens(SimpleNamespace(language_model=lm), pref.cuda(), beam, entry, temp, stop)
    outcome = "exact"
    for s in range(S):
        b = int(scores[s].argmax())
        mine = toks[s, b, : int(lens[s, b])].cpu().tolist()
        want = {}
        for name, rb in (("rb", True), ("exact", False)):
            ot, osc, ol, order = O.generate_beam_tokens(sd, pref[s:s + 1], n_head=n_head, n_layer=NL, beam_size=beam, entry_length=entry,
                                                        temperature=temp, stop_token=stop, rb=rb)
            want[name] = ot[order[0]][: int(ol[order[0]])].tolist()
        if mine == want["rb"]:
            continue
        if want["rb"] != want["exact"]:
            outcome = "unstable"          # the search itself flips under rounding: nothing to conclude
            continue
        # A margin below the bf16 noise of the product's own rounding instants (not the oracle's) also flips a search.  The same
        # network with fp16 operands carries 8x less rounding noise: if that run reproduces the oracle's caption, the bf16 difference
        # was a near-tie (the case that prompted this: first-step log-probabilities -1.8400 / -1.8461 in the oracle, -1.8443 / -1.8417
        # in the product), not a defect of the search.
        lm.set_precision(16)
        t16, s16, l16 = generate_beam_tokens(SimpleNamespace(language_model=lm), pref[s:s + 1].cuda(), beam, entry, temp, stop)
        lm.set_precision("bf16")
        b16 = int(s16[0].argmax())
        if t16[0, b16, : int(l16[0, b16])].cpu().tolist() == want["exact"]:
            outcome = "unstable"
            continue
        raise AssertionError((args, s, mine, want["rb"]))
    return outcome


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    master = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0, n, fails, unstable = time.time(), 0, [], 0
    while time.time() - t0 < budget:
        case = master.randrange(1 << 30)
        try:
            unstable += one(random.Random(case)) == "unstable"
            n += 1
        except Exception as e:
            fails.append(case)
            print("FAIL case", case, repr(e)[:500], flush=True)
    print(f"{n} searches in {time.time() - t0:.0f} s, {len(fails)} failures {fails}, {unstable} with a near-tie (the caption flips between the rounding-point and the exact oracle, or between the "
          f"bf16 and the fp16 product)")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
