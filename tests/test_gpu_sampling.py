"""cc_sample_step (on-device temperature / repetition penalty / top-k / top-p / multinomial) against the oracle restatements of the
reference's helpers (oracle.nucleus_final_p = inference/base.py:165-181, oracle.top_k_top_p_filtering = utils.py:5-30,
oracle.repetition_penalty_apply = utils.py:33-37, each pinned to the reference by tests/golden/filters.npz in test_oracle_golden)."""
import pytest
import torch

from oracle import clipcap_oracle as oracle

pytestmark = pytest.mark.gpu


def _eng():
    from clipcap_amd import engine
    return engine


def _margin_ok(probs_sorted_cum: torch.Tensor, top_p: float, tol: float = 2e-5) -> bool:
    """the reference's own fp32 cumsum decides the cut; skip rows where the crossing is within rounding of top_p"""
    return bool(((probs_sorted_cum - top_p).abs() > tol).all())


@pytest.mark.parametrize("V,scale", [(97, 3.0), (1000, 4.0), (50257, 6.0)])
@pytest.mark.parametrize("top_p,top_k,temperature", [(0.8, None, 1.0), (0.5, 40, 0.7), (0.95, 300, 1.3), (1.0, None, 1.0), (0.05, None, 1.0)])
def test_nucleus_distribution_matches_reference_semantics(V, scale, top_p, top_k, temperature):
    torch.manual_seed(V + int(top_p * 100))
    R = 7
    top_k = min(top_k, V) if top_k else None          # the reference's topk() raises beyond V; the kernel treats >= V as "all"
    logits = (torch.randn(R, V) * scale).cuda()
    u = torch.rand(R, device="cuda")
    nt, probs = _eng().sample_step(logits, u, temperature=temperature, top_k=top_k or 0, top_p=top_p, mode=0, return_probs=True)
    x = logits.cpu() / temperature
    ref = oracle.nucleus_final_p(x, top_p=top_p, top_k=top_k)
    p_sorted = torch.softmax(x, -1).sort(-1, descending=True).values
    if top_k:
        p_sorted = p_sorted[:, :top_k]
    cum = p_sorted.cumsum(-1)
    probs = probs.cpu()
    checked = 0
    for r in range(R):
        if top_p < 1.0 and not _margin_ok(cum[r], top_p):
            continue
        checked += 1
        assert torch.allclose(probs[r], ref[r], atol=2e-6, rtol=1e-4), (r, (probs[r] - ref[r]).abs().max())
        if top_p < 1.0:     # at 1.0 the reference's cut is wherever its fp32 cumsum first rounds up to 1: only ~1e-7 tails differ
            assert (probs[r] > 0).sum() == (ref[r] > 0).sum()
    assert checked >= R - 2
    assert torch.allclose(probs.sum(-1), torch.ones(R), atol=1e-5)
    # the drawn token is the inverse CDF (token-id order) of that distribution at u
    cdf = probs.double().cumsum(-1)
    for r in range(R):
        t = int(nt[r])
        lo = float(cdf[r, t - 1]) if t > 0 else 0.0
        hi = float(cdf[r, t])
        assert probs[r, t] > 0
        assert lo - 1e-6 <= float(u[r]) <= hi + 1e-6


@pytest.mark.parametrize("V", [97, 50257])
@pytest.mark.parametrize("top_p,top_k,temperature", [(0.9, 0, 1.0), (0.5, 10, 0.9), (0.0, 5, 1.0), (1.0, 0, 0.95)])
def test_filter_mode_matches_top_k_top_p_filtering(V, top_p, top_k, temperature):
    torch.manual_seed(V + top_k)
    R = 5
    logits = (torch.randn(R, V) * 5.0).cuda()
    u = torch.rand(R, device="cuda")
    nt, probs = _eng().sample_step(logits, u, temperature=temperature, top_k=top_k, top_p=top_p, mode=1, return_probs=True)
    probs = probs.cpu()
    checked = 0
    for r in range(R):
        x = logits[r].cpu() / temperature
        ref = torch.softmax(oracle.top_k_top_p_filtering(x.clone(), top_k=top_k, top_p=top_p), -1)
        xs = x.clone()
        if top_k > 0:
            xs[xs < xs.topk(top_k).values[-1]] = -float("inf")
        cum = torch.softmax(xs.sort(descending=True).values, -1).cumsum(-1)
        if 0 < top_p < 1.0 and not _margin_ok(cum, top_p):
            continue
        checked += 1
        assert torch.allclose(probs[r], ref, atol=2e-6, rtol=1e-4), (r, (probs[r] - ref).abs().max())
        if top_p < 1.0:
            assert (probs[r] > 0).sum() == (ref > 0).sum()
    assert checked >= R - 2


def test_repetition_penalty_and_history():
    torch.manual_seed(3)
    R, V = 4, 211
    logits = (torch.randn(R, V) * 2.0).cuda()
    hist = torch.randint(0, V, (R, 9), device="cuda")
    hist[:, 3] = hist[:, 1]                         # duplicates are penalised once
    u = torch.rand(R, device="cuda")
    _, probs = _eng().sample_step(logits, u, top_p=1.0, mode=0, history=hist, hist_len=6, repetition_penalty=1.2, return_probs=True)
    for r in range(R):
        x = oracle.repetition_penalty_apply(logits[r].cpu().clone(), hist[r, :6].cpu().unique(), 1.2)
        assert torch.allclose(probs[r].cpu(), torch.softmax(x, -1), atol=2e-6, rtol=1e-4)


def test_ties_at_the_threshold_are_kept_in_index_order():
    V = 64
    x = torch.zeros(1, V)
    x[0, 5] = 2.0
    x[0, [9, 20, 33, 47]] = 1.0                     # four equal runners-up
    u = torch.tensor([0.5], device="cuda")
    # nucleus, exactly k = 3: the top-1 plus the first two ties by index
    _, probs = _eng().sample_step(x.cuda(), u, top_k=3, top_p=1.0, mode=0, return_probs=True)
    assert sorted(torch.nonzero(probs[0].cpu()).flatten().tolist()) == [5, 9, 20]
    # filter mode keeps every tie of the k-th value (utils.py: logits < kth are removed)
    _, probs = _eng().sample_step(x.cuda(), u, top_k=3, top_p=0.0, mode=1, return_probs=True)
    assert sorted(torch.nonzero(probs[0].cpu()).flatten().tolist()) == [5, 9, 20, 33, 47]


def test_draws_follow_the_distribution():
    torch.manual_seed(11)
    V, N = 50, 4096
    row = torch.randn(V) * 2.0
    logits = row.expand(N, V).contiguous().cuda()
    u = torch.rand(N, device="cuda")
    nt, probs = _eng().sample_step(logits, u, top_p=0.9, mode=0, return_probs=True)
    freq = torch.bincount(nt.cpu().long(), minlength=V).double() / N
    assert (freq - probs[0].cpu().double()).abs().max() < 0.03
    assert freq[probs[0].cpu() == 0].sum() == 0


# ---------------------------------------------------------------------------------------------------------------------------
# The variants generate() uses (inference/no_beam.py, inference/nucleus_sampling.py) end to end on the KV-cached device loop,
# against the REFERENCE's per-step pre-sampling distributions (tests/golden/sampling_steps.npz: torch.multinomial patched to a
# forced token sequence with repeats, repetition penalty active from step 0 through the text prefix).
# ---------------------------------------------------------------------------------------------------------------------------

class _DotTokenizer:
    eos_token, bos_token = "<eos>", "<bos>"

    def __init__(self, dot):
        self.dot = dot

    def encode(self, s, return_tensors=None):
        return [self.dot]

    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)


@pytest.mark.parametrize("case", ["no_beam", "no_beam_topk", "nucleus"])
def test_generate_variants_step_distributions_vs_reference(case, monkeypatch):
    import numpy as np
    from types import SimpleNamespace
    from clipcap_amd import engine
    from clipcap_amd.inference import base, no_beam, nucleus_sampling
    from clipcap_amd.model.gpt2 import GPT2LM
    from tests.util import load_golden, sd_of
    g, b = load_golden("sampling_steps"), load_golden("gpt2_tiny")
    D, n_layer, n_head, V, npos = [int(v) for v in b["cfg"]]
    sd = {k: v for k, v in sd_of(b).items() if k != "lm_head.weight"}      # tied to wte: the scaled wte below must be the one that stays
    sd["transformer.wte.weight"] = sd["transformer.wte.weight"] * float(g["wte_scale"])
    lm = GPT2LM(n_embd=D, n_layer=n_layer, n_head=n_head, vocab_size=V, n_positions=npos)
    lm.load_state_dict(sd, strict=False)
    model = SimpleNamespace(language_model=lm.to("cuda"))
    top_p, top_k, temp, pen, stop = [float(v) for v in g[case + ".kw"]]
    forced = [int(t) for t in g[case + ".forced"]]
    head = torch.from_numpy(g[case + ".head"])
    rec, logit_rec = [], []
    real = engine.sample_step

    def spy(logits, u, **kw):       # record the product's distribution, then force the reference run's token
        _, probs = real(logits, u, return_probs=True, **kw)
        rec.append(probs[0].cpu())
        logit_rec.append((logits[0].cpu().clone(), kw["history"][0, : kw["hist_len"]].cpu().clone() if kw.get("history") is not None else None))
        return torch.full((logits.shape[0],), forced[len(rec) - 1], dtype=torch.int32, device=logits.device)

    monkeypatch.setattr(base, "sample_step", spy)
    pref = torch.from_numpy(g[case + ".prefix"]).cuda()
    tok = _DotTokenizer(int(stop))
    if case.startswith("no_beam"):
        text = no_beam.generate_no_beam(model, tok, pref, number_to_generate=1, text_prefix_tokens=head if head.numel() else None, top_p=top_p,
                                        top_k=top_k, entry_length=len(forced), temperature=temp, repetition_penalty=pen)
    else:
        text = nucleus_sampling.generate_nucleus_sampling(model, tok, pref, number_to_generate=1, entry_length=len(forced), top_p=top_p,
                                                          top_k=int(top_k), temperature=temp)
    assert [int(s) for s in text[0].split()] == [int(t) for t in g[case + ".text"]]
    assert len(rec) == len(forced)
    got = torch.stack(rec).numpy()
    ref = g[case + ".probs"]
    # (1) the sampling rule itself, exactly: the product's distribution == the oracle's restatement applied to the product's own logits
    for i, (lg, hist) in enumerate(logit_rec):
        if case.startswith("no_beam"):
            want = oracle.no_beam_step_distribution(lg, hist if hist is not None and hist.numel() else None, top_p=top_p, top_k=int(top_k),
                                                    temperature=temp, repetition_penalty=pen, stop_token=int(stop))
        else:
            want = oracle.nucleus_final_p((lg / temp).unsqueeze(0), top_p=top_p, top_k=(int(top_k) or None))[0]
        same_set = bool(((got[i] > 0) == (want.numpy() > 0)).all())
        assert same_set or np.abs(got[i] - want.numpy()).max() <= 2e-3, (case, i)       # cut within rounding of top_p: one token may differ
        if same_set:
            assert np.abs(got[i] - want.numpy()).max() <= 2e-6, (case, i)
    # (2) against the reference's fp32 run: bf16 GPT-2 logits move probabilities by <= 2e-2; the kept set may differ at the top_p cut
    err = np.abs(got - ref).max(axis=1)
    print(f"{case}: per-step max |p - reference p| = {np.array2string(err, precision=4)}")
    assert (err <= 3e-2).sum() >= len(forced) - 2 and np.median(err) <= 1e-2
    # the repetition penalty is visibly active: without it the distribution of a step with repeats in the history is different
    if case == "no_beam":
        lg, hist = logit_rec[3]
        nopen = oracle.no_beam_step_distribution(lg, hist, top_p=top_p, top_k=int(top_k), temperature=temp, repetition_penalty=1.0)
        assert np.abs(nopen.numpy() - got[3]).max() > 1e-3


@pytest.mark.parametrize("V,scale,top_p,top_k,temperature", [(1106, 1.5494166708685264, 0.211, None, 1.28), (41153, 3.2492567638160006, 0.4, None, 1.49)])
def test_selection_passes_bucket_an_element_identically(V, scale, top_p, top_k, temperature):
    """Regression (found by tools/fuzz_decode_steps.py): the threshold element of one row of these draws sits within 1e-7 of a boundary
    of the sampler's linear pre-buckets; with the temperature multiply contracted into the bucket computation at one call site and
    not at another, the element changed bucket between the histogram pass and the radix passes, no crossing was found and the row
    kept all V tokens instead of its 30 / 285-token nucleus."""
    test_nucleus_distribution_matches_reference_semantics(V, scale, top_p, top_k, temperature)


def test_sentence_length_penalty_fires_like_the_reference():
    """no_beam.py:55-60 / utils.py:40-51 on rows where the penalty FIRES (history tokens whose filtered logit equals float(stop id)):
    tests/golden/length_penalty.npz holds the reference's own per-step distributions for such rows (oracle/gen_golden.py (13))."""
    import numpy as np
    from tests.util import load_golden
    g = load_golden("length_penalty")
    n = len([k for k in g if k.endswith(".logits")])
    assert n >= 6
    for ci in range(n):
        stop, temp, rep, top_p, top_k, want_len, factor, fired = [float(v) for v in g[f"c{ci}.kw"]]
        assert fired >= 2
        lg = torch.from_numpy(g[f"c{ci}.logits"]).cuda().unsqueeze(0)
        hist = torch.from_numpy(g[f"c{ci}.hist"]).cuda().unsqueeze(0)
        hl = hist.shape[1]
        u = torch.tensor([0.37], device="cuda")
        ref = np.zeros(lg.shape[1], dtype=np.float32)
        ref[g[f"c{ci}.idx"]] = g[f"c{ci}.probs"]
        nt, probs = _eng().sample_step(lg, u, temperature=temp, top_k=int(top_k), top_p=top_p, mode=1, history=hist, hist_len=hl,
                                       repetition_penalty=rep, return_probs=True, length_penalty_stop=int(stop),
                                       length_penalty=(hl / want_len) * factor)
        got = probs[0].cpu().numpy()
        # the device sums probability mass in 2^-32 fixed point relative to the largest kept value: a kept token whose probability is
        # below ~2.3e-10 (the rows whose penalised value towers over the rest: the reference's fp32 softmax leaves denormals there)
        # carries weight 0.  The kept SET is compared where the reference's probability is representable.
        big = ref > 1e-8
        assert ((got > 0)[big]).all() and not (got[ref == 0] > 0).any(), ci
        assert np.abs(got - ref).max() <= 2e-6, (ci, np.abs(got - ref).max())
        assert got[int(nt[0])] > 0
        # and it differs from the step without the penalty (unless the factor makes it the identity)
        _, off = _eng().sample_step(lg, u, temperature=temp, top_k=int(top_k), top_p=top_p, mode=1, history=hist, hist_len=hl,
                                    repetition_penalty=rep, return_probs=True)
        if abs((hl / want_len) * factor - 1.0) > 1e-6:
            assert np.abs(off[0].cpu().numpy() - got).max() > 1e-4, ci
        # an empty history switches it off (the reference's `tokens is not None`)
        _, p0 = _eng().sample_step(lg, u, temperature=temp, top_k=int(top_k), top_p=top_p, mode=1, return_probs=True, length_penalty_stop=int(stop),
                                   length_penalty=3.0)
        _, p1 = _eng().sample_step(lg, u, temperature=temp, top_k=int(top_k), top_p=top_p, mode=1, return_probs=True)
        assert torch.equal(p0, p1)
