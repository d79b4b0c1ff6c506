"""cc_beam_step (the device-side beam update of generate_beam, reference inference/base.py:82-119) against the oracle's beam_update
at GPT-2's vocabulary size: first step, later steps with stopped beams, compile-time (1-5, 8) and run-time beam widths, a padded
leading dimension, and the all-ties case (lowest flat index wins; also the overflow path of the candidate list)."""
import pytest
import torch

from clipcap_amd.engine import beam_step
from oracle import clipcap_oracle as O

pytestmark = pytest.mark.gpu


def _oracle_step(lg, first, S, beam, temp, stop, scores, seql, stopped):
    V = lg.shape[1]
    nts, srcs = [], []
    for s in range(S):
        sl = slice(s * beam, (s + 1) * beam)
        if first:
            nt, src, sc, ln, hs = O.beam_update(lg[s * beam:s * beam + 1].clone(), None, seql[sl].clone(), stopped[sl].clone(), beam_size=beam,
                                                temperature=temp, stop_token=stop)
            src = torch.zeros(beam, dtype=torch.int64)
        else:
            nt, src, sc, ln, hs = O.beam_update(lg[sl].clone(), scores[sl].clone(), seql[sl].clone(), stopped[sl].clone(), beam_size=beam,
                                                temperature=temp, stop_token=stop)
        scores[sl], seql[sl], stopped[sl] = sc, ln, hs
        nts.append(nt)
        srcs.append(src)
    return torch.cat(nts), torch.cat(srcs)


@pytest.mark.parametrize("beam,V,ld", [(5, 50257, 50304), (3, 50257, 50257), (7, 4099, 4104), (8, 1001, 1001), (1, 50257, 50304)])
def test_beam_step_matches_oracle(beam, V, ld):
    torch.manual_seed(beam * 1000 + V)
    S, temp, stop = 6, 0.9, 17
    R = S * beam
    scores = torch.zeros(R, device="cuda")
    seql = torch.ones(R, device="cuda")
    stopped = torch.zeros(R, dtype=torch.uint8, device="cuda")
    o_scores, o_seql, o_stopped = torch.zeros(R), torch.ones(R), torch.zeros(R, dtype=torch.bool)
    for step in range(5):
        buf = torch.randn(R, ld, device="cuda") * 3.0
        if step >= 1:
            buf[::3, stop] += 25.0                     # some beams pick the stop token and freeze
        lg = buf[:, :V]
        nt, sr = beam_step(lg, S, beam, temp, step == 0, stop, scores, seql, stopped)
        ont, osr = _oracle_step(lg.cpu().float(), step == 0, S, beam, temp, stop, o_scores, o_seql, o_stopped)
        torch.cuda.synchronize()
        assert torch.equal(nt.cpu().long(), ont), step
        if step > 0:
            assert torch.equal(sr.cpu().long(), osr), step
        # the oracle follows the reference's fp32 `softmax(-1).log()`, which is itself only good to ~1e-5 (a random sweep found a 1.2e-5
        # row); the kernel's first-step scores are checked against fp64 below
        assert torch.allclose(scores.cpu(), o_scores, rtol=1e-5, atol=3e-5), step
        if step == 0:
            exact = torch.log_softmax(lg.double().cpu()[::beam] / temp, -1).topk(beam, -1).values.reshape(-1)
            assert (scores.cpu().double() - exact).abs().max().item() <= 2e-6
        assert torch.equal(seql.cpu(), o_seql) and torch.equal(stopped.cpu().bool(), o_stopped), step
    assert o_stopped.any()


@pytest.mark.parametrize("beam", [5, 6])
def test_beam_step_all_ties_lowest_flat_index(beam):
    S, V = 3, 50257
    R = S * beam
    scores = torch.zeros(R, device="cuda")
    seql = torch.ones(R, device="cuda")
    stopped = torch.zeros(R, dtype=torch.uint8, device="cuda")
    lg = torch.zeros(R, V, device="cuda")
    nt, sr = beam_step(lg, S, beam, 1.0, True, 50256, scores, seql, stopped)
    assert nt.view(S, beam).cpu().tolist() == [list(range(beam))] * S
    nt, sr = beam_step(lg, S, beam, 1.0, False, 50256, scores, seql, stopped)
    assert nt.view(S, beam).cpu().tolist() == [list(range(beam))] * S and sr.view(S, beam).cpu().tolist() == [[0] * beam] * S


def test_beam_advance_matches_indexing():
    """cc_beam_advance = tokens[src] ++ next, wte[next], ancestry rows gathered (base.py:104-117), against plain tensor indexing."""
    from clipcap_amd.engine import DecodeSession
    from clipcap_amd.model.gpt2 import GPT2LM
    torch.manual_seed(3)
    lm = GPT2LM(n_embd=64, n_layer=1, n_head=4, vocab_size=211, n_positions=32).to("cuda")
    wte = lm.get_input_embeddings().weight.detach()
    S, beam, n = 3, 4, 9
    R = S * beam
    sess = DecodeSession(lm.engine, R, 20)
    sess.forward(torch.randn(R, 6, 64, device="cuda"))
    sess.row_map[:, :6] = torch.randint(0, R, (R, 6), dtype=torch.int32, device="cuda")
    before = sess.row_map.clone()
    tin = torch.randint(0, 211, (R, n), dtype=torch.int32, device="cuda")
    tout = torch.full((R, n), -7, dtype=torch.int32, device="cuda")
    nxt = torch.randint(0, 211, (R,), dtype=torch.int32, device="cuda")
    src = torch.randint(0, beam, (R,), dtype=torch.int32, device="cuda")
    x = torch.empty(R, 1, 64, device="cuda")
    sess.beam_advance(beam, nxt, src, wte, 5, tin, tout, x)
    g = ((torch.arange(R, device="cuda") // beam) * beam + src).long()
    assert torch.equal(tout[:, :5], tin[g, :5]) and torch.equal(tout[:, 5], nxt) and (tout[:, 6:] == -7).all()
    assert torch.equal(x.view(R, 64), wte[nxt.long()])
    assert torch.equal(sess.row_map[:, :6], before[g, :6])
    assert torch.equal(sess.row_map[:, 6:], torch.arange(R, dtype=torch.int32, device="cuda").view(R, 1).expand(R, 14))


def _partials(lg, V):
    """what the decode lm_head epilogue writes (gemm.hip.h EpiLogits): per (row, 64-column block) max and sum of exp(x - max)"""
    R = lg.shape[0]
    npart = (V + 63) // 64
    pad = torch.full((R, npart * 64), float("-inf"), device=lg.device)
    pad[:, :V] = lg[:, :V]
    blk = pad.view(R, npart, 64)
    pmax = blk.max(dim=2).values
    psum = torch.exp(blk - pmax[:, :, None]).sum(dim=2)
    return (torch.cat((pmax.reshape(-1), psum.reshape(-1))).contiguous(), npart)


@pytest.mark.parametrize("beam,V,ld", [(5, 50257, 50304), (3, 50257, 50257), (8, 1001, 1001), (1, 50257, 50304), (4, 130, 136)])
def test_beam_step_from_partials_matches_oracle(beam, V, ld):
    """cc_beam_step_p: the one-launch update driven by the lm_head epilogue's partials (temperature 1) — same tokens, source rows,
    scores, lengths and stop flags as the oracle, through stopped beams."""
    torch.manual_seed(beam * 77 + V)
    S, stop = 6, 17
    R = S * beam
    scores = torch.zeros(R, device="cuda")
    seql = torch.ones(R, device="cuda")
    stopped = torch.zeros(R, dtype=torch.uint8, device="cuda")
    o_scores, o_seql, o_stopped = torch.zeros(R), torch.ones(R), torch.zeros(R, dtype=torch.bool)
    for step in range(5):
        buf = torch.randn(R, ld, device="cuda") * 3.0
        if step >= 1:
            buf[::3, stop] += 25.0
        lg = buf[:, :V]
        nt, sr = beam_step(lg, S, beam, 1.0, step == 0, stop, scores, seql, stopped, None, _partials(lg, V))
        ont, osr = _oracle_step(lg.cpu().float(), step == 0, S, beam, 1.0, stop, o_scores, o_seql, o_stopped)
        torch.cuda.synchronize()
        assert torch.equal(nt.cpu().long(), ont), step
        if step > 0:
            assert torch.equal(sr.cpu().long(), osr), step
        assert torch.allclose(scores.cpu(), o_scores, rtol=1e-5, atol=3e-5), step
        assert torch.equal(seql.cpu(), o_seql) and torch.equal(stopped.cpu().bool(), o_stopped), step
    assert o_stopped.any()


def test_beam_step_from_partials_all_ties_takes_the_exact_overflow_path():
    S, V, beam = 3, 50257, 5
    R = S * beam
    scores = torch.zeros(R, device="cuda")
    seql = torch.ones(R, device="cuda")
    stopped = torch.zeros(R, dtype=torch.uint8, device="cuda")
    lg = torch.zeros(R, V, device="cuda")
    part = _partials(lg, V)
    nt, sr = beam_step(lg, S, beam, 1.0, True, 50256, scores, seql, stopped, None, part)
    assert nt.view(S, beam).cpu().tolist() == [list(range(beam))] * S
    nt, sr = beam_step(lg, S, beam, 1.0, False, 50256, scores, seql, stopped, None, part)
    assert nt.view(S, beam).cpu().tolist() == [list(range(beam))] * S and sr.view(S, beam).cpu().tolist() == [[0] * beam] * S
