"""Mapper alone: per-tensor gradient error vs the exact (fp64) oracle for a random dout — is the top layer's MLP branch special?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from clipcap_amd.engine import MapperEngine
from oracle import clipcap_oracle as O

torch.manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
E, D, P, L, H, N, B = 24, 256, 6, 6, 4, 2, 4
prec = 16 if (len(sys.argv) > 2 and sys.argv[2] == "fp16") else None
eng = MapperEngine(E, D, L, P, H, N, device="cuda", precision=prec)
sd = {}
for k, v in eng.views(eng.arena.w32).items():
    if "norm" in k and k.endswith("weight"):
        t = 1.0 + 0.05 * torch.randn(v.shape)
    elif k.endswith(".bias"):
        t = 0.02 * torch.randn(v.shape)
    elif "prefix_const" in k:
        t = torch.randn(v.shape)
    else:
        t = torch.randn(v.shape) * 0.5 / v.shape[-1] ** 0.5
    sd[k] = t
    v.copy_(t)
x = torch.randn(B, E)
out = eng.forward(x.cuda(), save=True)
dout = torch.randn_like(out) * 1e-2
eng.arena.grads().zero_()
eng.backward(dout)
got = {k: v.cpu().double() for k, v in eng.views(eng.arena.g32).items()}
for name, rb in (("rb", "fp16" if prec else True), ("exact", False)):
    sdr = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.mapper_forward(sdr, x.double(), projection_length=P, num_heads=H, num_layers=N, rb=rb)
    (ref * dout.cpu().double()).sum().backward()
    print(name, "out err", (out.cpu().double() - ref.detach()).abs().max().item())
    for k in sd:
        r = ((got[k] - sdr[k].grad).norm() / sdr[k].grad.norm()).item()
        print(f"  {k:45s} {r:.2e}")
