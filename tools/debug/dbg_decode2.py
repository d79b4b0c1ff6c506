import sys, torch
sys.path.insert(0, '.')
from types import SimpleNamespace
from tests.test_gpu_configs import _medium_lm
from clipcap_amd.inference.base import generate_beam_tokens
lm, _ = _medium_lm(24)
model = SimpleNamespace(language_model=lm)
gen = torch.Generator(device="cuda").manual_seed(9)
pref = torch.randn(64, 10, 1024, generator=gen, device="cuda") * 0.5
toks, scores, lens = generate_beam_tokens(model, pref, 5, 12, 1.0, 50256)
bad = 0
for i in range(64):
    t1, s1, l1 = generate_beam_tokens(model, pref[i:i+1], 5, 12, 1.0, 50256)
    b, b1 = int(scores[i].argmax()), int(s1[0].argmax())
    same = torch.equal(toks[i, b], t1[0, b1])
    if not same:
        bad += 1
        print(i, "batched", toks[i, b].tolist(), "%.4f" % float(scores[i, b]), "alone", t1[0, b1].tolist(), "%.4f" % float(s1[0, b1]),
              "batched top2 gap %.4f" % float(scores[i].sort(descending=True)[0][:2].diff().abs()), "alone top2 gap %.4f" % float(s1[0].sort(descending=True)[0][:2].diff().abs()))
print("mismatches", bad, "of 64")
