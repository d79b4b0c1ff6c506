import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import test_gpu_sampling as TS
from oracle import clipcap_oracle as oracle
for (V, scale, top_p, top_k, temperature, rows) in [(1106, 1.5494166708685264, 0.211, None, 1.28, [6]), (41153, 3.2492567638160006, 0.4, None, 1.49, [1])]:
    torch.manual_seed(V + int(top_p * 100))
    R = 7
    logits = (torch.randn(R, V) * scale).cuda()
    u = torch.rand(R, device="cuda")
    nt, probs = TS._eng().sample_step(logits, u, temperature=temperature, top_k=top_k or 0, top_p=top_p, mode=0, return_probs=True)
    x = logits.cpu() / temperature
    ref = oracle.nucleus_final_p(x, top_p=top_p, top_k=top_k)
    ps = torch.softmax(x.double(), -1).sort(-1, descending=True).values
    ps32 = torch.softmax(x, -1).sort(-1, descending=True).values
    probs = probs.cpu()
    for r in rows:
        nk, nr = int((probs[r] > 0).sum()), int((ref[r] > 0).sum())
        cum64 = ps[r].cumsum(-1); cum32 = ps32[r].cumsum(-1)
        print(V, r, "kernel kept", nk, "ref kept", nr, "cum64 around", [f"{v:.7f}" for v in cum64[max(0,min(nk,nr)-2):max(nk,nr)+2].tolist()],
              "cum32", [f"{v:.7f}" for v in cum32[max(0,min(nk,nr)-2):max(nk,nr)+2].tolist()], "top_p", top_p)
