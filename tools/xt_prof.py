#!/usr/bin/env python
"""Per-phase profile of the XCD-team decode engine (decode_xt.hip, CC_XT_PROF=1): time per phase and time spent polling, per layer, from the
I/O wave's s_memrealtime stamps of every workgroup."""
import os
import sys

import torch

os.environ["CC_XT_PROF"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clipcap_amd import _lib  # noqa: E402
from clipcap_amd.engine import DecodeSession  # noqa: E402
from clipcap_amd.model.gpt2 import GPT2LM  # noqa: E402

NL = int(os.environ.get("NL", 24))
_lib.lib().cc_decode_mode(_lib.lib().cc_decode_mode(-1) | 4)
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
assert _lib.lib().cc_decode_last_path() == 2
print(f"last step (pos {sess.pos - 1}): {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us for the whole cc_decode_fwd_x call, {NL} layers")
ws = sess._ws[1]
off = 0
for n in (R * D * 4, R * D * 4, R * D * 2, R * 3 * D * 2, R * D * 2, R * 4 * D * 2, R * D * 2, R * 4, R * 4, R * 4, 8 * R * 4 * D * 4, (7 * 8 + 8) * 4, 256 * 21 * 8,
          (16 + 8 * 32 + 16) * 4):
    off = ((off + 255) & ~255) + n
off = (off + 255) & ~255
prof = ws[off:off + 256 * 32 * 8].view(torch.int64).view(256, 32).double().cpu() / 100.0
names = ["P1 c_attn (ln_1 panel)", "P2 attention", "P3 attn.c_proj (DMA panel)", "P4 c_fc (ln_2 panel, gelu)", "P5 mlp.c_proj (chunked DMA)"]
sub = ["poll", "panel fill", "K loop (MFMA waves)", "epilogue+drain+arrive"]
tot = 0.0
for ph in range(5):
    parts = [prof[:, 4 * ph + k].mean().item() / NL for k in range(4)]
    tot += sum(parts)
    if ph == 1:
        print(f"{names[ph]:30s}: {sum(parts):6.2f} us per layer = poll {parts[0]:5.2f} | items {parts[1]:5.2f} | arrive {parts[3]:5.2f}")
    else:
        print(f"{names[ph]:30s}: {sum(parts):6.2f} us per layer = " + " | ".join(f"{sub[k]} {parts[k]:5.2f}" for k in range(4)))
print(f"ln_f (once): {prof[:, 20].mean().item():.2f} us;  sum of the phase means: {tot:.1f} us per layer = {tot * NL:.0f} us per position")
