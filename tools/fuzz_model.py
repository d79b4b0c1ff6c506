#!/usr/bin/env python
"""Randomised whole-path parity sweep: random (legal) mapper / GPT-2 geometries, batch sizes, ragged captions, operand type and forced
GEMM tile; one training step on the GPU against the CPU oracle with the kernels' rounding points (the check of tests/test_gpu_edges.py),
and KV-cached decode logits against the full forward of the same engine.  Not part of the test suite:
    python tools/fuzz_model.py [seconds] [seed]"""
import os
import random
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clipcap_amd import _lib
from tests import test_gpu_edges as E


LAST = {}


def one_step(rng):
    hd_m = rng.choice([8, 16, 24, 32, 64, 96])
    H = rng.choice([1, 2, 4])
    D = hd_m * H
    if D % 8:
        return None
    n_head = rng.choice([h for h in (1, 2, 3, 4, 6, 8) if D % h == 0 and (D // h) % 8 == 0])
    P, L, N, n_layer = rng.randint(1, 6), rng.randint(1, 6), rng.randint(1, 2), rng.randint(1, 2)
    E_ = 8 * rng.randint(1, 12)
    V = rng.randint(50, 1500)
    B, cap = rng.randint(1, 7), rng.randint(1, 14)
    prec = rng.choice([None, 16, 32])
    mode = rng.choice([-1, -1, 0, 3, 4, 5, 6, 7])
    if "FUZZ_TILE" in os.environ:
        mode = int(os.environ["FUZZ_TILE"])
    if "FUZZ_PREC" in os.environ:
        prec = {"bf16": None, "16": 16, "32": 32}[os.environ["FUZZ_PREC"]]
    args = dict(E=E_, D=D, P=P, L=L, H=H, N=N, n_head=n_head, n_layer=n_layer, V=V, B=B, cap=cap, prec=prec, tile=mode)
    LAST.update(args=args)
    eng, sd, cfg = E._build(E_, D, P, L, H, N, n_head, n_layer, V, L + cap + 2, seed=rng.randint(0, 999), prec=prec)
    torch.manual_seed(rng.randint(0, 1 << 30))
    tokens = torch.randint(1, V, (B, cap))
    for b in range(B):
        if rng.random() < 0.4:
            tokens[b, rng.randint(0, cap):] = -1
    if rng.random() < 0.2:
        tokens[rng.randint(0, B - 1), rng.randint(0, cap - 1)] = 0
    embeds = torch.randn(B, E_)
    old = _lib.lib().cc_gemm_tile_mode(mode)
    try:
        try:
            if VERBOSE:
                classify(eng, sd, cfg, tokens, embeds)
            E._check(eng, sd, cfg, tokens, embeds, tol=4e-3)
        except AssertionError as e:
            # outside the fixed tolerances: is the kernel further from the like-for-like oracle than that oracle is from exact arithmetic?
            # (gradients of LayerNorm weights at a dozen rows are sums with heavy cancellation: the 16-bit rounding points alone move them)
            why = classify(eng, sd, cfg, tokens, embeds)
            if why is not None:
                raise AssertionError(f"{e} | {why}")
            NOISE.append(str(e)[:60])
    finally:
        _lib.lib().cc_gemm_tile_mode(old)
    return args


NOISE = []
VERBOSE = []


def classify(eng, sd, cfg, tokens, embeds):
    from oracle import clipcap_oracle as O
    fp16 = eng.scaler is not None
    eng.zero_grad()
    loss = float(eng.forward_backward(tokens.cuda(), embeds.cuda()))
    unscale = 1.0 / float(eng.scaler.scale) if fp16 else 1.0
    got = {k: v.cpu().double() * unscale for k, v in eng.mapper.views(eng.mapper.arena.g32).items()}
    res = {}
    x3 = eng.mapper.op_dtype == 2
    for name, rb in (("rb", "bf16x3" if x3 else ("fp16" if fp16 else True)), ("exact", False)):
        sdr = {k: v.double().clone().requires_grad_(k.startswith("transformer_mapper.")) for k, v in sd.items()}
        l = O.clipcap_loss(sdr, tokens, embeds.double(), cfg=cfg, rb=rb)
        l.backward()
        res[name] = (float(l), {k[len("transformer_mapper."):]: v.grad for k, v in sdr.items() if v.grad is not None})
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    bad = []
    dl_k, dl_o = abs(loss - res["rb"][0]), abs(res["rb"][0] - res["exact"][0])
    if dl_k > max(4e-3, 1e-3 * abs(res["rb"][0])) and dl_k > 3.0 * dl_o + 1e-3:      # a mean over a handful of targets: 1e-3 relative
        bad.append(f"loss: kernel-oracle {dl_k:.2e}, oracle-exact {dl_o:.2e}")
    for k, g in got.items():
        ek, eo = rel(g, res["rb"][1][k]), rel(res["rb"][1][k], res["exact"][1][k])
        # The gradients behind the MLP's ReLU (fc1, norm2) see relu'(h) = [h > 0]: wherever |pre-activation| is below the forward
        # rounding noise the kernel's and the oracle's masks differ, and a flipped element is wrong by its full size — the error goes
        # like sqrt(fraction flipped), 4-14 % at a dozen rows in bf16, in BOTH operand types' own units:
        # 3.5-4.8 % on every layer's fc1 / norm2 in bf16 against ~1 % elsewhere, 0.8-2 % against 0.3 % in fp16).  Not a kernel property.
        lim = 0.25 if (".mlp.fc1." in k or ".norm2." in k) else 6e-2
        if ek > lim and ek > 3.0 * eo:
            bad.append(f"{k}: kernel-oracle {ek:.2e}, oracle-exact {eo:.2e}")
    if VERBOSE:
        print(f"loss kernel {loss:.6f} oracle(rb) {res['rb'][0]:.6f} exact {res['exact'][0]:.6f}")
        for k, g in got.items():
            print(f"  {k:45s} |g| {g.norm().item():.3e}  kernel-oracle {rel(g, res['rb'][1][k]):.2e}  kernel-exact {rel(g, res['exact'][1][k]):.2e}  oracle-exact "
                  f"{rel(res['rb'][1][k], res['exact'][1][k]):.2e}")
    return "; ".join(bad) if bad else None


def one_full(rng):
    """full finetune (GPT-2 gradients too), half of the cases with the three dropout sites active: the kernels' counter-based masks are
    read back through cc_dropout_mask and handed to the oracle (the check of tests/test_gpu_dropout.py)"""
    from oracle import clipcap_oracle as O
    from tests import test_gpu_dropout as TD
    hd = rng.choice([16, 32, 64, 64])            # 64: the MFMA attention kernels (the only ones with attention dropout)
    n_head = rng.choice([1, 2, 3])
    D = hd * n_head
    H = rng.choice([h for h in (1, 2, 4) if D % h == 0 and (D // h) % 8 == 0])
    P, L, N, n_layer = rng.randint(1, 4), rng.randint(1, 4), rng.randint(1, 2), rng.randint(1, 2)
    E_, V, B, cap = 8 * rng.randint(1, 8), rng.randint(60, 900), rng.randint(1, 5), rng.randint(1, 10)
    use_drop = hd == 64 and rng.random() < 0.6
    mode = rng.choice([-1, -1, 0, 3, 4, 5, 6, 7])
    args = dict(full=True, E=E_, D=D, P=P, L=L, H=H, N=N, n_head=n_head, n_layer=n_layer, V=V, B=B, cap=cap, drop=use_drop, tile=mode)
    LAST.update(args=args)
    eng, sd, cfg = TD._build(E_, D, P, L, H, N, n_head, n_layer, V, L + cap + 2, seed=rng.randint(0, 999))
    torch.manual_seed(rng.randint(0, 1 << 30))
    tokens, embeds = torch.randint(1, V, (B, cap)), torch.randn(B, E_)
    if rng.random() < 0.4:
        tokens[rng.randint(0, B - 1), rng.randint(0, cap):] = -1
    T = L + cap
    drop = None
    kw = {}
    if use_drop:
        p_e, p_a, p_r, seed = round(rng.uniform(0.05, 0.3), 2), round(rng.uniform(0.05, 0.3), 2), round(rng.uniform(0.05, 0.3), 2), rng.randrange(1 << 40)
        kw = dict(dropout=(p_e, p_a, p_r, seed))
        drop = {"p_embd": p_e, "p_attn": p_a, "p_resid": p_r, "embd": TD._mask(seed, 0, 0, p_e, (B, T, D)),
                "attn": [TD._mask(seed, 1, l, p_a, (B, n_head, T, T)) for l in range(n_layer)],
                "resid_attn": [TD._mask(seed, 2, l, p_r, (B, T, D)) for l in range(n_layer)],
                "resid_mlp": [TD._mask(seed, 3, l, p_r, (B, T, D)) for l in range(n_layer)]}
    old = _lib.lib().cc_gemm_tile_mode(mode)
    try:
        loss = float(eng.forward_backward(tokens.cuda(), embeds.cuda(), **kw))
    finally:
        _lib.lib().cc_gemm_tile_mode(old)
    kept = int((tokens > 0).sum())
    if kept == 0:
        assert loss == 0.0
        return args
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.clipcap_loss(sdr, tokens, embeds, cfg=cfg, rb=True, drop=drop)
    ref.backward()
    if abs(loss - float(ref.detach())) > max(4e-3, 1e-3 * abs(float(ref.detach()))):
        # a mean over a handful of targets: the same yardstick as for the gradients — the like-for-like oracle's own distance from exact arithmetic
        with torch.no_grad():
            le = float(O.clipcap_loss({kk: vv.double() for kk, vv in sd.items()}, tokens, embeds.double(), cfg=cfg, rb=False,
                                      drop=None if drop is None else {dk: ([m.double() for m in dv] if isinstance(dv, list) else (dv.double() if torch.is_tensor(dv) else dv))
                                                                      for dk, dv in drop.items()}))
        if VERBOSE:
            print(f"  loss: kernel {loss:.6f} oracle {float(ref.detach()):.6f} exact {le:.6f}")
        assert abs(loss - float(ref.detach())) <= 3.0 * abs(float(ref.detach()) - le) + 1e-3, (loss, float(ref.detach()), le, args)
        NOISE.append(f"loss {loss:.4f} vs {float(ref.detach()):.4f} (exact {le:.4f})")
    exact = None
    for pre, e in (("transformer_mapper.", eng.mapper), ("language_model.", eng.gpt2)):
        for k, v in e.views(e.arena.g32).items():
            r = sdr[pre + k].grad
            if "lm_head" in k or r is None:
                continue
            err = ((v.cpu() - r).norm() / r.norm().clamp_min(1e-12)).item()
            relu = (".mlp.fc1." in k or ".norm2." in k) and pre.startswith("transformer_mapper")      # ReLU-mask flips, see classify(); one flip weighs more the fewer rows there are
            lim = (0.4 if B * (P + L) <= 8 else 0.25) if relu else 8e-2
            if err > lim:
                # outside the fixed tolerance: is the kernel further from the like-for-like oracle than that oracle is from exact (fp64,
                # unrounded) arithmetic?  Gradients that are differences of nearly equal terms (attention queries / keys over two or
                # three positions, LayerNorm weights at a handful of rows) move by tens of per cent under the 16-bit rounding points alone.
                if exact is None:
                    sde = {kk: vv.double().clone().requires_grad_(True) for kk, vv in sd.items()}
                    O.clipcap_loss(sde, tokens, embeds.double(), cfg=cfg, rb=False,
                                   drop=None if drop is None else {dk: ([m.double() for m in dv] if isinstance(dv, list) else (dv.double() if torch.is_tensor(dv) else dv))
                                                                   for dk, dv in drop.items()}).backward()
                    exact = {kk: vv.grad for kk, vv in sde.items() if vv.grad is not None}
                eo = ((r.double() - exact[pre + k]).norm() / exact[pre + k].norm().clamp_min(1e-30)).item()
                if VERBOSE:
                    print(f"  {pre + k}: kernel-oracle {err:.2e}, oracle-exact {eo:.2e}")
                assert err <= 3.0 * eo, (pre + k, err, eo)
                NOISE.append(f"{k}: {err:.2e} vs oracle-exact {eo:.2e}")
    return args


def one_decode(rng):
    from clipcap_amd.engine import DecodeSession
    from clipcap_amd.model.gpt2 import GPT2LM
    hd = rng.choice([8, 16, 32, 64])
    n_head = rng.choice([1, 2, 4, 12])
    D = hd * n_head
    R, T0, extra = rng.randint(1, 40), rng.randint(1, 12), rng.randint(1, 4)
    V = rng.randint(60, 2000)
    args = dict(D=D, n_head=n_head, R=R, T0=T0, extra=extra, V=V)
    LAST.update(args=args)
    torch.manual_seed(rng.randint(0, 999))
    lm = GPT2LM(n_embd=D, n_layer=rng.randint(1, 2), n_head=n_head, vocab_size=V, n_positions=32).to("cuda")
    x = torch.randn(R, T0 + extra, D, device="cuda") * 0.4
    sess = DecodeSession(lm.engine, R, T0 + extra)
    last = sess.forward(x[:, :T0])
    for t in range(extra):
        last = sess.forward(x[:, T0 + t:T0 + t + 1])
    full = lm.engine.logits(x)[:, -1, :V]
    d = (last[:, :V] - full).float()
    scale = max(1.0, full.abs().max().item())
    assert d.abs().max().item() <= 1e-2 * scale and d.pow(2).mean().sqrt().item() <= 3e-3 * scale, (d.abs().max().item(), scale)
    return args


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--case":          # re-run one case by its seed, verbosely
        rng = random.Random(int(sys.argv[2]))
        x = rng.random()
        fn = one_step if x < 0.55 else (one_full if x < 0.8 else one_decode)
        VERBOSE.append(1)
        print(fn.__name__, fn(rng), "ok")
        return
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    master = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0, n, fails = time.time(), 0, []
    while time.time() - t0 < budget:
        case = master.randrange(1 << 30)
        rng = random.Random(case)
        x = rng.random()
        fn = one_step if x < 0.55 else (one_full if x < 0.8 else one_decode)
        try:
            if fn(rng) is not None:
                n += 1
        except Exception as e:
            fails.append(case)
            print("FAIL case", case, fn.__name__, LAST.get("args"), repr(e)[:300], flush=True)
    print(f"{n} cases in {time.time() - t0:.0f} s, {len(fails)} failures {fails}; {len(NOISE)} cases outside the test suite's fixed tolerances but accounted for by "
          f"rounding (classify(): small-batch loss means, ReLU-mask flips, oracle-vs-exact distance)")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
