import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import test_gpu_sampling as TS
from oracle import clipcap_oracle as oracle
V, top_p, top_k, temperature = 124, 0.0, 0, 1.19
torch.manual_seed(V + top_k)
R = 5
logits = (torch.randn(R, V) * 5.0).cuda()
u = torch.rand(R, device="cuda")
nt, probs = TS._eng().sample_step(logits, u, temperature=temperature, top_k=top_k, top_p=top_p, mode=1, return_probs=True)
probs = probs.cpu()
for r in range(R):
    x = logits[r].cpu() / temperature
    ref = torch.softmax(oracle.top_k_top_p_filtering(x.clone(), top_k=top_k, top_p=top_p), -1)
    print(r, "max abs diff", (probs[r] - ref).abs().max().item(), "nonzero", int((probs[r] > 0).sum()), int((ref > 0).sum()), "sum", probs[r].sum().item())
