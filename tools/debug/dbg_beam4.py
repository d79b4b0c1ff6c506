import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from clipcap_amd.engine import beam_step
from tests import test_gpu_beam as TB
beam, V, ld = 9, 191, 238
torch.manual_seed(beam * 1000 + V)
S, temp, stop = 6, 0.9, 17
R = S * beam
scores = torch.zeros(R, device="cuda"); seql = torch.ones(R, device="cuda"); stopped = torch.zeros(R, dtype=torch.uint8, device="cuda")
o_scores, o_seql, o_stopped = torch.zeros(R), torch.ones(R), torch.zeros(R, dtype=torch.bool)
for step in range(5):
    buf = torch.randn(R, ld, device="cuda") * 3.0
    if step >= 1:
        buf[::3, stop] += 25.0
    lg = buf[:, :V]
    nt, sr = beam_step(lg, S, beam, temp, step == 0, stop, scores, seql, stopped)
    ont, osr = TB._oracle_step(lg.cpu().float(), step == 0, S, beam, temp, stop, o_scores, o_seql, o_stopped)
    torch.cuda.synchronize()
    ok = torch.equal(nt.cpu().long(), ont)
    print("step", step, "tokens equal", ok)
    if not ok:
        d = (nt.cpu().long() != ont).nonzero().flatten().tolist()
        for i in d:
            s = i // beam
            print("  row", i, "sample", s, "kernel tok/src", int(nt[i]), int(sr[i]), "oracle tok/src", int(ont[i]), int(osr[i]),
                  "kernel score", float(scores[i]), "oracle score", float(o_scores[i]), "stopped(o)", o_stopped[s*beam:(s+1)*beam].tolist())
        break
