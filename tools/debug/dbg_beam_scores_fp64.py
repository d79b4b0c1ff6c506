import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from clipcap_amd.engine import beam_step
from tests import test_gpu_beam as TB
beam, V, ld = 1, 46465, 46465
torch.manual_seed(beam * 1000 + V)
S, temp, stop = 6, 0.9, 17
R = S * beam
scores = torch.zeros(R, device="cuda"); seql = torch.ones(R, device="cuda"); stopped = torch.zeros(R, dtype=torch.uint8, device="cuda")
o_scores, o_seql, o_stopped = torch.zeros(R), torch.ones(R), torch.zeros(R, dtype=torch.bool)
buf = torch.randn(R, ld, device="cuda") * 3.0
lg = buf[:, :V]
nt, sr = beam_step(lg, S, beam, temp, True, stop, scores, seql, stopped)
ont, osr = TB._oracle_step(lg.cpu().float(), True, S, beam, temp, stop, o_scores, o_seql, o_stopped)
print("kernel", scores.cpu().tolist()); print("oracle", o_scores.tolist()); print("nt", nt.cpu().tolist(), ont.tolist(), "stop", stop)
lp = torch.log_softmax(lg.double().cpu() / temp, -1)
print("fp64  ", lp.max(-1).values.tolist())
