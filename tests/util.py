import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def sd_of(g, prefix="sd."):
    return {k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith(prefix)}


def seeded_full_model(g):
    """State dict (reference key names, torch tensors) + oracle cfg of a config2_full / config4_full fixture, regenerated from the
    seed (tests/seeded.py) and checked against the checksum the generator stored."""
    from tests import seeded
    E, D, P, L, H, N, n_head, NL, V, NPOS, seed, full = [int(v) for v in g["cfg"]]
    gsd = seeded.state_dict(seeded.gpt2_shapes(D, NL, V, NPOS), seed)
    msd = seeded.state_dict(seeded.mapper_shapes(E, D, P, L, N), seed + 1)
    chk = np.concatenate([seeded.checksum(gsd), seeded.checksum(msd)])
    assert np.array_equal(chk, g["param_checksum"]), "seeded parameters differ from the ones the fixture was generated with"
    sd = {"language_model." + k: torch.from_numpy(v) for k, v in gsd.items()}
    sd.update({"transformer_mapper." + k: torch.from_numpy(v) for k, v in msd.items()})
    cfg = dict(projection_length=P, prefix_length=L, heads=H, layers=N, n_head=n_head, n_layer=NL)
    dims = dict(E=E, D=D, P=P, L=L, H=H, N=N, n_head=n_head, NL=NL, V=V, NPOS=NPOS, full=bool(full))
    return sd, cfg, dims


def sampled(t):
    """(norm, strided sample) of a tensor, the form the large-gradient entries of the fixtures are stored in."""
    from tests.seeded import sample_idx
    a = t.detach().cpu().numpy().reshape(-1)
    return float(np.sqrt((a.astype(np.float64) ** 2).sum())), a[sample_idx(a.size)]
