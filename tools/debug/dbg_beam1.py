import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from clipcap_amd.engine import beam_step
for beam in (1, 2):
  for V in (46465, 46464, 46466, 50257, 1001, 4099, 30001):
    torch.manual_seed(beam * 1000 + V)
    S, temp, stop = 6, 0.9, 17
    R = S * beam
    scores = torch.zeros(R, device="cuda"); seql = torch.ones(R, device="cuda"); stopped = torch.zeros(R, dtype=torch.uint8, device="cuda")
    buf = torch.randn(R, V, device="cuda") * 3.0
    nt, sr = beam_step(buf, S, beam, temp, True, stop, scores, seql, stopped)
    lp = torch.log_softmax(buf.double().cpu() / temp, -1)
    ref = lp[::beam].topk(beam, -1)
    print(beam, V, "score err", (scores.cpu().double().view(S, beam) - ref.values).abs().max().item(), "tok ok", torch.equal(nt.cpu().view(S, beam).long(), ref.indices))
