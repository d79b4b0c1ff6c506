"""Re-run one tools/fuzz_beam_search.py case and print the first-step log-probabilities of the tokens the product and the oracle chose."""
import os, random, sys
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from clipcap_amd.engine import DecodeSession
from clipcap_amd.model.gpt2 import GPT2LM
from oracle import clipcap_oracle as O
case, sample = int(sys.argv[1]), int(sys.argv[2])
rng = random.Random(case)
hd = rng.choice([16, 32, 64]); n_head = rng.choice([1, 2, 4]); D, NL = hd * n_head, rng.randint(1, 2); V = rng.randint(50, 400)
S, L0, beam, entry = rng.randint(1, 4), rng.randint(1, 6), rng.randint(1, 8), rng.randint(3, 12)
temp, stop = round(rng.uniform(0.7, 1.3), 2), rng.randrange(V)
torch.manual_seed(rng.randrange(1 << 30))
lm = GPT2LM(n_embd=D, n_layer=NL, n_head=n_head, vocab_size=V, n_positions=32).to("cuda")
with torch.no_grad():
    for n_, p_ in lm.named_parameters():
        if "wte" in n_:
            p_.mul_(8.0)
sd = {"language_model." + k: v.detach().cpu().float() for k, v in lm.state_dict().items() if "lm_head" not in k}
pref = torch.randn(S, L0, D) * 0.7
print(dict(D=D, n_head=n_head, NL=NL, V=V, S=S, L0=L0, beam=beam, entry=entry, temp=temp, stop=stop))
sess = DecodeSession(lm.engine, S, L0 + entry)
lg = sess.forward(pref.cuda())[sample, :V].cpu().double() / temp
lp = torch.log_softmax(lg, -1)
for name, rb in (("rb", True), ("exact", False)):
    o = O.gpt2_logits(sd, pref[sample:sample + 1], n_head, NL, pre="language_model.", rb=rb)[0, -1].double() / temp
    olp = torch.log_softmax(o, -1)
    top = olp.topk(4)
    print(name, "oracle top4", top.indices.tolist(), [f"{v:.5f}" for v in top.values.tolist()], "| product at those", [f"{lp[i]:.5f}" for i in top.indices.tolist()])
top = lp.topk(4)
print("product top4", top.indices.tolist(), [f"{v:.5f}" for v in top.values.tolist()])
