#!/usr/bin/env python
"""Where does the batched split-bf16 beam decode of tests/test_gpu_x3.py::test_x3_batched_beam_equals_per_sample_all_64 part from the
per-sample decode of the same prefix, and by how much did the competing candidates differ there?  (A flip is legitimate only at a
near-tie: the batched and the single-sample steps run different GEMM tilings, i.e. different fp32 summation orders.)
usage: x3_beam_divergence.py [sample ...]"""
import os
import sys
from types import SimpleNamespace

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch  # noqa: E402

from clipcap_amd.inference.base import generate_beam_tokens  # noqa: E402
from tests.test_gpu_configs import _medium_lm  # noqa: E402

if __name__ == "__main__":
    samples = [int(a) for a in sys.argv[1:]] or [47]
    lm, _ = _medium_lm(24, precision=32)
    model = SimpleNamespace(language_model=lm)
    gen = torch.Generator(device="cuda").manual_seed(9)
    pref = torch.randn(64, 10, 1024, generator=gen, device="cuda") * 0.5
    for i in samples:
        for n in range(1, 13):
            tb, sb, lb = generate_beam_tokens(model, pref, 5, n, 1.0, 50256)
            ta, sa, la = generate_beam_tokens(model, pref[i:i + 1], 5, n, 1.0, 50256)
            kb = sorted((tuple(tb[i, b].tolist()), float(sb[i, b] * lb[i, b])) for b in range(5))
            ka = sorted((tuple(ta[0, b].tolist()), float(sa[0, b] * la[0, b])) for b in range(5))
            same = [x[0] for x in kb] == [x[0] for x in ka]
            dmax = max(abs(x[1] - y[1]) for x, y in zip(kb, ka)) if same else float("nan")
            print(f"sample {i} after {n:2d} steps: beam sets {'equal' if same else 'DIFFER'}; max |sum-logprob batched - alone| = {dmax:.2e}")
            if not same:
                only_b = [x for x in kb if x[0] not in [y[0] for y in ka]]
                only_a = [x for x in ka if x[0] not in [y[0] for y in kb]]
                for x in only_b:
                    print(f"   only batched: ...{x[0][-3:]} sum-logprob {x[1]:.6f}")
                for x in only_a:
                    print(f"   only alone  : ...{x[0][-3:]} sum-logprob {x[1]:.6f}")
                print(f"   batched 5 sum-logprobs: {sorted(round(x[1], 6) for x in kb)}")
                print(f"   alone   5 sum-logprobs: {sorted(round(x[1], 6) for x in ka)}")
                break
