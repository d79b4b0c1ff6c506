#!/usr/bin/env python
"""How much of a beam group's KV history is shared (what k_decode_attn_group can skip): per generated position, the number of distinct
(cache row, position) pairs the 5 ancestry tables of a caption name, against the 5 x ctx rows the per-row kernel reads.
    python tools/decode_union_stats.py            # bench.py --mode decode geometry: 64 prefixes, beam 5, GPT-2-medium random init"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipcap_amd.engine import DecodeSession, beam_buffers, beam_step  # noqa: E402
from clipcap_amd.model.gpt2 import GPT2LM  # noqa: E402


@torch.no_grad()
def main():
    torch.manual_seed(1234)
    dev = "cuda"
    lm = GPT2LM(n_embd=1024, n_layer=int(os.environ.get("NL", 24)), n_head=16, vocab_size=50257, n_positions=1024).to(dev)
    g = lm.engine
    S, L0, beam, entry = 64, 10, 5, 67
    R = S * beam
    V = g.dims["V"]
    embeds = torch.randn(S, L0, 1024, device=dev) * 0.5
    wte = lm.get_input_embeddings().weight.detach()
    scores = torch.zeros(R, device=dev)
    seq_lengths = torch.ones(R, device=dev)
    has_stopped = torch.zeros(R, dtype=torch.uint8, device=dev)
    base = (torch.arange(S, device=dev, dtype=torch.int32) * beam).repeat_interleave(beam)
    sess = DecodeSession(g, S, L0 + entry)
    lg = torch.empty(R, V, device=dev)
    lg[::beam] = sess.forward(embeds)
    bufs = beam_buffers(dev, S, beam, V)
    next_tok, src = beam_step(lg, S, beam, 1.0, True, 50256, scores, seq_lengths, has_stopped, bufs)
    sess = sess.expand((base // beam).to(torch.int32), R)
    tok = [torch.zeros(R, entry, dtype=torch.int32, device=dev) for _ in range(2)]
    x = torch.empty(R, 1, 1024, device=dev)
    sess.beam_advance(beam, next_tok, None, wte, 0, tok[1], tok[0], x)
    tot_u = tot_rows = 0
    hist = []
    for step in range(1, entry):
        ctx = sess.pos
        rm = sess.row_map[:, :ctx].view(S, beam, ctx).cpu()
        nu = sum(int(torch.unique(rm[s, :, j]).numel()) for s in range(S) for j in range(ctx)) / S + beam
        hist.append((ctx + 1, nu))
        tot_u += nu
        tot_rows += beam * (ctx + 1)
        logits = sess.forward(x, partials=True, group=beam)
        next_tok, src = beam_step(logits, S, beam, 1.0, False, 50256, scores, seq_lengths, has_stopped, bufs, sess.lpart)
        sess.beam_advance(beam, next_tok, src, wte, step, tok[(step - 1) & 1], tok[step & 1], x)
    print("ctx -> mean distinct rows per group (of 5 x ctx):", [(c, round(u, 1)) for c, u in hist[::6]])
    print(f"mean union size {tot_u / len(hist):.1f} entries vs {tot_rows / len(hist):.1f} rows read per group by the per-row kernel "
          f"({tot_rows / tot_u:.2f}x fewer K/V rows)")


if __name__ == "__main__":
    main()
