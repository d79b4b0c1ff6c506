#!/usr/bin/env python
"""Randomised sweep of the device-side decode bookkeeping against the oracle: cc_beam_step (beam widths 1..10, vocabularies 50..60000,
padded leading dimensions, stop tokens, frozen beams) and cc_sample_step / cc_sample_step_lp (nucleus / top-k / temperature, both filter
conventions, repetition and sentence-length penalties) —
the assertions of tests/test_gpu_beam.py and tests/test_gpu_sampling.py on random parameters.  Not part of the test suite:
    python tools/fuzz_decode_steps.py [seconds] [seed]"""
import os
import random
import sys
import time
import traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_beam as TB
from tests import test_gpu_sampling as TS


def beam_case(beam, V, ld):
    """tests/test_gpu_beam.py::test_beam_step_matches_oracle, except that a step whose kept candidates are the oracle's in another ORDER
    is accepted when the candidates' average scores agree to 3e-5 (the device's expf / logf differ from the host's by an ulp and the
    reference's fp32 softmax().log() is itself good to ~1e-5; two candidates tied that closely sort either way) — the run stops there, the states having diverged."""
    import torch
    from clipcap_amd.engine import beam_step
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
        if torch.equal(nt.cpu().long(), ont) and (step == 0 or torch.equal(sr.cpu().long(), osr)):
            assert torch.allclose(scores.cpu(), o_scores, rtol=1e-5, atol=3e-5), step
            continue
        for s in range(S):
            sl = slice(s * beam, (s + 1) * beam)
            a = sorted(zip(nt[sl].tolist(), (sr[sl].tolist() if step else [0] * beam), (scores[sl] / seql[sl]).tolist()))
            b = sorted(zip(ont[sl].tolist(), (osr[sl].tolist() if step else [0] * beam), (o_scores[sl] / o_seql[sl]).tolist()))
            assert [x[:2] for x in a] == [x[:2] for x in b], (step, s, "different candidates")
            assert all(abs(x[2] - y[2]) <= 3e-5 for x, y in zip(a, b)), (step, s)      # the oracle's fp32 softmax().log() is good to ~1e-5
        return "tie"
    return "exact"


def length_penalty_case(V, stop, temp, rep, top_p, top_k, want_len, factor, hl, seed):
    """The no_beam step with the sentence-length penalty (cc_sample_step_lp) on rows whose logits sit on a 0.25 grid, so that history tokens
    hit float(stop) exactly now and then (the reference's value comparison, utils.py:40-51), against the oracle's restatement of
    no_beam.py:45-63 (pinned by tests/golden/length_penalty.npz)."""
    import numpy as np
    import torch
    from clipcap_amd.engine import sample_step
    from oracle import clipcap_oracle as O
    gen = torch.Generator().manual_seed(seed)
    lg = (torch.randn(V, generator=gen) * 3.0 + stop * 0.6).mul(4).round().div(4) * temp     # grid values; / temp restores them exactly for temp = 2^k
    hist = torch.randint(0, V, (hl,), generator=gen)
    for t in hist[: max(1, hl // 3)].tolist():
        lg[t] = float(stop) * temp
    pen = (hl / want_len) * factor
    _, probs = sample_step(lg.cuda().unsqueeze(0), torch.tensor([0.5], device="cuda"), temperature=temp, top_k=top_k, top_p=top_p, mode=1,
                           history=hist.cuda().unsqueeze(0), hist_len=hl, repetition_penalty=rep, return_probs=True, length_penalty_stop=stop,
                           length_penalty=pen)
    ref = O.no_beam_step_distribution(lg, hist, top_p=top_p, top_k=top_k, temperature=temp, repetition_penalty=rep, stop_token=stop,
                                      desired_sentence_length=want_len, sentence_length_factor=factor).numpy()
    got = probs[0].cpu().numpy()
    fired = int(((lg[hist] / temp) == stop).sum())
    if np.array_equal(got > 1e-8, ref > 1e-8):
        assert np.abs(got - ref).max() <= 3e-6, np.abs(got - ref).max()
        return fired, "exact"
    # The rows sit on a grid, so the top-k / top-p cut often falls INSIDE a group of equal values: the reference keeps whichever members its
    # sort put first, the device the first in index order.  Then: the kept sets before the penalty must hold the same VALUES,
    # and the device's distribution must be the penalty + softmax applied to ITS kept set.
    _, p0 = sample_step(lg.cuda().unsqueeze(0), torch.tensor([0.5], device="cuda"), temperature=temp, top_k=top_k, top_p=top_p, mode=1,
                        history=hist.cuda().unsqueeze(0), hist_len=hl, repetition_penalty=rep, return_probs=True)
    r0 = O.no_beam_step_distribution(lg, hist, top_p=top_p, top_k=top_k, temperature=temp, repetition_penalty=rep).numpy()
    x = O.repetition_penalty_apply(lg.clone(), hist, rep) if rep != 1.0 else lg.clone()
    x = (x / temp).numpy()
    kd, kr = p0[0].cpu().numpy() > 0, r0 > 0
    big = x >= x.max() - 18.0                                    # 2^-32 fixed point: compare the members that can carry weight
    if not np.array_equal(np.sort(x[kd & big]), np.sort(x[kr & big])):
        # the cumulative mass at the cut is within fp32 rounding of top_p: one side keeps the boundary value's group (or one more member
        # of it), the other does not — every token the two sets disagree on must sit at the cut, i.e. at or below the smallest value both keep
        both = kd & kr & big
        diff = (kd ^ kr) & big
        assert both.any() and x[diff].max() <= x[both].min() + 1e-6 and np.unique(x[diff]).size <= 2, "kept sets differ away from the cut"
    inh = np.zeros(V, dtype=bool)
    inh[hist.numpy()] = True
    y = np.where(kd, np.where(inh & (x == np.float32(stop)), x * np.float32(pen), x), -np.inf).astype(np.float64)
    e = np.exp(y - y.max())
    assert np.abs(got - e / e.sum()).max() <= 3e-6
    return fired, "cut"


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0, n, fails, skipped, saturated, ties, lp_fired, lp_cut = time.time(), 0, [], 0, 0, 0, 0, 0
    while time.time() - t0 < budget:
        kind = rng.choice(["beam", "nucleus", "filter", "length"])
        V = rng.choice([rng.randint(50, 400), rng.randint(400, 6000), rng.randint(6000, 60000)])
        try:
            if kind == "beam":
                args = (rng.randint(1, 10), V, V + rng.choice([0, 0, 3, 8, 47]))
                if beam_case(*args) == "tie":
                    ties += 1
            elif kind == "length":
                stop = rng.choice([5, 13, min(V - 1, 40)])
                args = (V, stop, rng.choice([1.0, 0.5, 2.0]), rng.choice([1.0, 1.0, 1.25, 2.0]), rng.choice([0.95, 0.9, 0.7, 0.0]), rng.choice([0, 0, 8, 40]),
                        rng.choice([5, 10, 50]), rng.choice([0.0, 0.5, 1.0, 3.0]), rng.randint(1, 40), rng.randint(0, 1 << 30))
                if args[4] == 0.0 and args[5] == 0:
                    continue                                  # no filter at all: see the 2^-32 note below
                f, how = length_penalty_case(*args)
                lp_fired += f > 0
                lp_cut += how == "cut"
            elif kind == "nucleus":
                k = rng.choice([None, None, rng.randint(1, 500)])
                args = (V, rng.uniform(1.0, 7.0), rng.choice([1.0, round(rng.uniform(0.02, 0.999), 3)]), k, round(rng.uniform(0.5, 1.5), 2))
                TS.test_nucleus_distribution_matches_reference_semantics(*args)
            else:
                top_k = rng.choice([0, rng.randint(1, min(V, 300))])
                # no filter at all (top_p = 0 and top_k = 0) is left out: the sampler's weights are 2^-32 fixed point, so a token more than
                # 22 nats below the row maximum gets probability 0 instead of < 2.3e-10, which the nonzero-count assertion would flag
                top_p = rng.choice([1.0, round(rng.uniform(0.05, 0.99), 3)] + ([0.0] if 0 < top_k <= 20 else []))     # same reason: the k-th token must stay above 2^-32
                args = (V, top_p, top_k, round(rng.uniform(0.5, 1.5), 2))
                TS.test_filter_mode_matches_top_k_top_p_filtering(*args)
        except Exception as e:
            tb = traceback.format_exc()
            if "assert checked >= R - 2" in tb:      # too many rows of this draw had their nucleus cut within rounding of top_p: nothing was compared
                skipped += 1
                continue
            if "(probs[r] > 0).sum() == (ref[r] > 0).sum()" in tb:
                # a row whose top probability rounds to 1.0 in fp32: the reference's fp32 cumsum saturates there, `cum <= cut` then keeps
                # EVERY token (each with < 1e-8 of the mass); the kernel's integer masses keep the nucleus.  The probabilities agree to
                # atol 2e-6 (asserted just before that line); only the count of non-zero entries differs.
                saturated += 1
                continue
            fails.append((kind, args))
            print("FAIL", kind, args, repr(e)[:300], flush=True)
        n += 1
    print(f"{n} cases in {time.time() - t0:.0f} s, {len(fails)} failures, {skipped} draws skipped (nucleus cut within rounding of top_p), {saturated} with an fp32-saturated reference cumsum, {ties} beam steps with a tie below the reference's own rounding sorted the other way; sentence-length penalty: fired in {lp_fired} rows, {lp_cut} rows compared by mass (top_p cut inside a tie)")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
