"""CPU emulation of sm_select pass 0 + radix passes for one row, to see where the selection is lost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import test_gpu_sampling as TS
V, scale, top_p, temperature, row = 1106, 1.5494166708685264, 0.211, 1.28, 6
torch.manual_seed(V + int(top_p * 100))
R = 7
logits = (torch.randn(R, V) * scale).cuda()
u = torch.rand(R, device="cuda")
# which rows fail, and does the failure persist when the row is alone / repeated?
for trial in range(3):
    nt, probs = TS._eng().sample_step(logits, u, temperature=temperature, top_k=0, top_p=top_p, mode=0, return_probs=True)
    print("trial", trial, "kept per row", [(int((probs[r] > 0).sum())) for r in range(R)])
one = logits[row:row + 1].contiguous()
nt, probs = TS._eng().sample_step(one, u[row:row + 1].contiguous(), temperature=temperature, top_k=0, top_p=top_p, mode=0, return_probs=True)
print("alone kept", int((probs[0] > 0).sum()))
x = logits[row].cpu().numpy().astype(np.float32)
inv_temp = np.float32(1.0 / temperature)
v = (x * inv_temp).astype(np.float32)
m, mn = v.max(), v.min()
xscale = np.float32((np.float32(2048) - np.float32(0.001)) / (m - mn))
lb = np.minimum(np.maximum(((v - mn).astype(np.float32) * xscale).astype(np.float32), 0), 2047).astype(np.int64)
w = (np.exp((v - m).astype(np.float32)).astype(np.float32) * np.float32(4294967040.0)).astype(np.uint64)
Z = int(w.sum()); target = int(float(np.float32(top_p)) * float(Z))
print("Z", Z, "target", target, "m", m, "mn", mn, "xscale", xscale, "lb range", lb.min(), lb.max())
order = np.argsort(-v, kind="stable")
cum = np.cumsum(w[order].astype(np.float64))
n_keep = int(np.searchsorted(cum, target, side="left")) + 1
print("expected kept", n_keep, "threshold value", v[order[n_keep - 1]], "bucket", lb[order[n_keep - 1]], "count in that bucket", int((lb == lb[order[n_keep - 1]]).sum()))
