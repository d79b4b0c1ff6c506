"""Seeded parameter sets for the shape-faithful fixtures (TEST INFRASTRUCTURE; shared by oracle/gen_golden.py and tests/).

A config-2 / config-4 model has 0.17-0.4 G parameters: storing them next to the reference's outputs would put hundreds of MB
in the repository.  Instead the parameters are a pure numpy function of (tensor name, shape, seed) — numpy's PCG64 +
``standard_normal`` / ``uniform`` streams — so ``oracle/gen_golden.py`` (which feeds them to the REFERENCE) and the tests
(which feed them to the oracle and to the HIP path) build bit-identical state dicts, and the fixture only holds inputs, the
reference's outputs (sub-sampled where they are large) and a checksum of the parameters that guards the regeneration.

Distributions follow the defaults the reference gets: torch.nn.Linear U(+-1/sqrt(fan_in)) for weight and bias,
prefix_const / pos_embeddings N(0,1), GPT-2 N(0, 0.02) with residual projections scaled by 1/sqrt(2 n_layer) — except that
biases and LayerNorm affine parameters are perturbed (HF zero / identity init would leave them unexercised).
"""
from __future__ import annotations

import zlib
from typing import Dict, List, Tuple

import numpy as np

MAPPER_LAYER = [("norm1.weight", "ln_w"), ("norm1.bias", "ln_b"), ("attn.to_queries.weight", "lin_w"), ("attn.to_keys_values.weight", "lin_w"),
                ("attn.project.weight", "lin_w"), ("attn.project.bias", "lin_b"), ("norm2.weight", "ln_w"), ("norm2.bias", "ln_b"),
                ("mlp.fc1.weight", "lin_w"), ("mlp.fc1.bias", "lin_b"), ("mlp.fc2.weight", "lin_w"), ("mlp.fc2.bias", "lin_b")]


def mapper_shapes(E: int, D: int, P: int, L: int, N: int, W: int = 1, use_pos: bool = False) -> List[Tuple[str, tuple, str]]:
    """(name, shape, kind) in the reference's state-dict order (clipcap/model/mapper.py:113-160)."""
    Hm = int(D * 2.0)
    out = [("transformer.layers.%d." % i + n, s, k) for i in range(N) for (n, k), s in zip(MAPPER_LAYER, [
        (D,), (D,), (D, D), (2 * D, D), (D, D), (D,), (D,), (D,), (Hm, D), (Hm,), (D, Hm), (D,)])]
    head = [("linear.weight", (P * D, E), "lin_w"), ("linear.bias", (P * D,), "lin_b:%d" % E), ("prefix_const", (L, D), "unit")]
    if W > 1 and use_pos:
        head.append(("pos_embeddings", (W * P, D), "unit"))
    return head + out


def gpt2_shapes(D: int, NL: int, V: int, NPOS: int) -> List[Tuple[str, tuple, str]]:
    """HF GPT2LMHeadModel state-dict names (lm_head.weight is tied to wte and not listed)."""
    out = [("transformer.wte.weight", (V, D), "gpt_w"), ("transformer.wpe.weight", (NPOS, D), "gpt_w")]
    for i in range(NL):
        pre = "transformer.h.%d." % i
        out += [(pre + "ln_1.weight", (D,), "ln_w"), (pre + "ln_1.bias", (D,), "ln_b"),
                (pre + "attn.c_attn.weight", (D, 3 * D), "gpt_w"), (pre + "attn.c_attn.bias", (3 * D,), "gpt_b"),
                (pre + "attn.c_proj.weight", (D, D), "gpt_proj:%d" % NL), (pre + "attn.c_proj.bias", (D,), "gpt_b"),
                (pre + "ln_2.weight", (D,), "ln_w"), (pre + "ln_2.bias", (D,), "ln_b"),
                (pre + "mlp.c_fc.weight", (D, 4 * D), "gpt_w"), (pre + "mlp.c_fc.bias", (4 * D,), "gpt_b"),
                (pre + "mlp.c_proj.weight", (4 * D, D), "gpt_proj:%d" % NL), (pre + "mlp.c_proj.bias", (D,), "gpt_b")]
    out += [("transformer.ln_f.weight", (D,), "ln_w"), ("transformer.ln_f.bias", (D,), "ln_b")]
    return out


def _tensor(name: str, shape: tuple, kind: str, seed: int) -> np.ndarray:
    rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
    arg = None
    if ":" in kind:
        kind, arg = kind.split(":")
    if kind == "lin_w":
        b = 1.0 / np.sqrt(shape[1])
        return rng.uniform(-b, b, size=shape).astype(np.float32)
    if kind == "lin_b":       # bound from the layer's fan_in: passed explicitly, or (square-ish layers) recovered by the caller
        b = 1.0 / np.sqrt(float(arg)) if arg else 0.03
        return rng.uniform(-b, b, size=shape).astype(np.float32)
    if kind == "unit":
        return rng.standard_normal(shape, dtype=np.float32)
    if kind == "ln_w":
        return (1.0 + 0.05 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
    if kind == "ln_b":
        return (0.05 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
    if kind == "gpt_w":
        return (0.02 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
    if kind == "gpt_b":
        return (0.02 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
    if kind == "gpt_proj":
        return ((0.02 / np.sqrt(2.0 * int(arg))) * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
    raise ValueError(kind)


def state_dict(shapes: List[Tuple[str, tuple, str]], seed: int, prefix: str = "") -> Dict[str, np.ndarray]:
    return {prefix + n: _tensor(n, s, k, seed) for n, s, k in shapes}


def checksum(sd: Dict[str, np.ndarray]) -> np.ndarray:
    """[crc32 of all bytes in name order, float64 sum of |values|]: guards regeneration on another numpy build."""
    crc, tot = 0, 0.0
    for k in sorted(sd):
        a = np.ascontiguousarray(sd[k])
        crc = zlib.crc32(a.tobytes(), crc)
        tot += float(np.abs(a.astype(np.float64)).sum())
    return np.array([float(crc), tot])


def sample_idx(n: int, k: int = 4096) -> np.ndarray:
    """Deterministic sub-sample of a flat tensor of n elements: all of it if small, else k evenly strided positions."""
    if n <= k:
        return np.arange(n)
    return (np.arange(k, dtype=np.int64) * n) // k
