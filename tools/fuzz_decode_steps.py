#!/usr/bin/env python
"""Randomised sweep of the device-side decode bookkeeping against the oracle: cc_beam_step (beam widths 1..10, vocabularies 50..60000,
padded leading dimensions, stop tokens, frozen beams) and cc_sample_step (nucleus / top-k / temperature, both filter conventions) —
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


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0, n, fails, skipped, saturated, ties = time.time(), 0, [], 0, 0, 0
    while time.time() - t0 < budget:
        kind = rng.choice(["beam", "nucleus", "filter"])
        V = rng.choice([rng.randint(50, 400), rng.randint(400, 6000), rng.randint(6000, 60000)])
        try:
            if kind == "beam":
                args = (rng.randint(1, 10), V, V + rng.choice([0, 0, 3, 8, 47]))
                if beam_case(*args) == "tie":
                    ties += 1
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
    print(f"{n} cases in {time.time() - t0:.0f} s, {len(fails)} failures, {skipped} draws skipped (nucleus cut within rounding of top_p), {saturated} with an fp32-saturated reference cumsum, {ties} beam steps with a tie below the reference's own rounding sorted the other way")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
