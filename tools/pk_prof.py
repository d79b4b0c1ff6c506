#!/usr/bin/env python
"""Per-phase profile of the persistent decode-layer launch (decode_pk.hip, CC_PK_PROF=1): poll time and total time per task, per phase."""
import os
import sys

import torch

os.environ["CC_PK_PROF"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clipcap_amd import _lib  # noqa: E402
from clipcap_amd.engine import DecodeSession  # noqa: E402
from clipcap_amd.model.gpt2 import GPT2LM  # noqa: E402

NL = int(os.environ.get("NL", 24))
_lib.lib().cc_decode_mode(3)                     # group attention + persistent layer launch
torch.manual_seed(1234)
lm = GPT2LM(n_embd=1024, n_layer=NL, n_head=16, vocab_size=50257, n_positions=1024).to("cuda")
ge = lm.engine
S, G, L0, D = 64, 5, 10, 1024
R = S * G
pref = torch.randn(S, L0, D, device="cuda") * 0.5
base = torch.arange(S, device="cuda", dtype=torch.int32).repeat_interleave(G)
s0 = DecodeSession(ge, S, 80)
s0.forward(pref)
sess = s0.expand(base, R)
x = torch.randn(R, 1, D, device="cuda") * 0.5
for t in range(int(os.environ.get("STEPS", 20))):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    sess.forward(x, partials=True, group=G)
    ev[1].record()
torch.cuda.synchronize()
sess.check()
print(f"last step (pos {sess.pos - 1}): {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us for the whole cc_decode_fwd_g call, {NL} layers")
ws = sess._ws[1]
off = 0
for n in (R * D * 4, R * D * 4, R * D * 2, R * 3 * D * 2, R * D * 2, R * 4 * D * 2, R * D * 2, R * 4, R * 4, R * 4, 8 * R * 4 * D * 4):
    off = ((off + 255) & ~255) + n
off = (off + 255) & ~255            # pk_ctr
off += (7 * 8 + 8) * 4
off = (off + 255) & ~255            # pk_prof
prof = ws[off:off + 256 * 21 * 8].view(torch.int64).view(256, 7, 3).double().cpu()
names = ["P1 c_attn", "P2 attention", "P3 c_proj", "P3f finish+ln_2", "P4 c_fc", "P5 mlp c_proj", "P5f finish+ln_1"]
tot = 0.0
for ph in range(7):
    n = prof[:, ph, 2]
    busy = n > 0
    wait = prof[busy, ph, 0].sum() / n[busy].sum() / 100.0
    total = prof[busy, ph, 1].sum() / n[busy].sum() / 100.0
    per_layer = prof[:, ph, 1].max() / 100.0 / NL
    tot += per_layer
    print(f"{names[ph]:18s}: {int(busy.sum()):3d} workgroups, {n[busy].mean() / NL:.2f} tasks each per layer; per task: poll {wait:6.2f} us, total {total:6.2f} us "
          f"(work {total - wait:5.2f}); busiest workgroup {per_layer:6.2f} us per layer")
print(f"sum over phases of the busiest workgroup's time per layer: {tot:.1f} us")
