import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import test_gpu_sampling as TS
from oracle import clipcap_oracle as oracle
for (V, scale, top_p, top_k, temperature) in [(161, 6.89713224467348, 0.643, None, 0.5), (5411, 6.3481825106439, 0.97, None, 0.52)]:
    torch.manual_seed(V + int(top_p * 100))
    R = 7
    logits = (torch.randn(R, V) * scale).cuda()
    u = torch.rand(R, device="cuda")
    nt, probs = TS._eng().sample_step(logits, u, temperature=temperature, top_k=top_k or 0, top_p=top_p, mode=0, return_probs=True)
    x = logits.cpu() / temperature
    ref = oracle.nucleus_final_p(x, top_p=top_p, top_k=top_k)
    ps = torch.softmax(x.double(), -1).sort(-1, descending=True).values
    probs = probs.cpu()
    for r in range(R):
        nk, nr = int((probs[r] > 0).sum()), int((ref[r] > 0).sum())
        print(V, r, "kernel kept", nk, "ref kept", nr, "top probs", [f"{v:.3e}" for v in ps[r, :4].tolist()], "cum", [f"{v:.7f}" for v in ps[r].cumsum(-1)[:4].tolist()],
              "maxdiff", (probs[r] - ref[r]).abs().max().item())
