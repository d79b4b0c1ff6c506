import sys, torch
sys.path.insert(0, '.')
from types import SimpleNamespace
from tests.test_gpu_configs import _medium_lm
from clipcap_amd.engine import DecodeSession
from clipcap_amd.inference.base import generate_beam_tokens
NL = int(sys.argv[1]) if len(sys.argv) > 1 else 24
lm, _ = _medium_lm(NL)
ge = lm.engine
torch.manual_seed(2)
x = torch.randn(320, 14, 1024, device="cuda") * 0.3
full = ge.logits(x[:5])
for R in (5, 64, 320):
    sess = DecodeSession(ge, R, 32)
    l = sess.forward(x[:R, :10]).clone()
    errs = [float((l[:5] - full[:, 9]).abs().max())]
    for t in range(10, 14):
        l = sess.forward(x[:R, t:t + 1])
        errs.append(float((l[:5] - full[:, t]).abs().max()))
    print("R", R, "max err vs reforward per step", ["%.3e" % e for e in errs], "scale", float(full.abs().max()))
model = SimpleNamespace(language_model=lm)
gen = torch.Generator(device="cuda").manual_seed(9)
pref = torch.randn(64, 10, 1024, generator=gen, device="cuda") * 0.5
for S in (64, 8, 2):
    toks, scores, lens = generate_beam_tokens(model, pref[:S], 5, 12, 1.0, 50256)
    print("S", S, "sample0 scores", scores[0].tolist(), "best", toks[0, int(scores[0].argmax())].tolist())
t1, s1, l1 = generate_beam_tokens(model, pref[:1], 5, 12, 1.0, 50256)
print("alone scores", s1[0].tolist(), "best", t1[0, int(s1[0].argmax())].tolist())
