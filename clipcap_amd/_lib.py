"""ctypes binding of libclipcap_hip.so (C ABI: include/clipcap_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a kernel call fails, this raises.
Build the library with ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C clipcap_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
# CLIPCAP_HIP_LIB selects the library: unset = the product build; "lab" = libclipcap_hip_lab.so (`make -C clipcap_amd/csrc lab`: the product plus the
# experiment kernels and environment A/B switches, tests/lab_*.py); anything else = a path.
_SEL = os.environ.get("CLIPCAP_HIP_LIB", "")
LIB_PATH = os.path.join(HERE, "libclipcap_hip.so") if not _SEL else os.path.join(HERE, "libclipcap_hip_lab.so") if _SEL == "lab" else _SEL
IS_LAB = _SEL == "lab"

ERR = {0: "ok", -1: "invalid argument", -2: "unsupported shape / alignment", -3: "kernel launch failed", -4: "invalid state"}


class HipExtensionMissing(RuntimeError):
    pass


class CCError(RuntimeError):
    pass


OP_BF16, OP_FP16, OP_X3 = 0, 1, 2      # CC_OP_BF16 / CC_OP_FP16 / CC_OP_BF16X3: operand mode of a model (cfg.op_dtype)


def op_dtype_of(precision) -> int:
    """--fp-precision of the reference (clipcap/train/args.py:30-34, handed to pl.Trainer at train.py:82) -> operand mode:

    * 32 (the reference's default) / 64 / "32" / "bf16x3": split bf16 operands — every GEMM as three bf16 MFMA terms
      (hi*hi + hi*lo + lo*hi), fp32 activations and attention: logits within 1e-3 of the fp32 reference at full depth, about a third
      of the GEMM rate.  (gfx950's fp32 MFMA, v_mfma_f32_32x32x2_f32, runs at 1/16 of the bf16 rate — three bf16 terms are ~5x
      faster than it; 64 runs the same mode: ~16 mantissa bits per operand, fp32 accumulation.)
    * 16 / "fp16": IEEE fp16 operands + dynamic loss scaling (what Lightning gives the reference for --fp-precision 16).
    * "bf16" (or None, the engines' own default = BASELINE.json's bf16 configurations): bf16 operands, the throughput mode.

    Accumulation, master weights, residual streams, LayerNorm statistics, loss and optimizer are fp32 in all three."""
    if precision in ("bf16", None) or (precision == OP_BF16 and not isinstance(precision, bool)):
        return OP_BF16
    if precision in (16, "fp16", "16", OP_FP16):
        return OP_FP16
    if precision in (32, 64, "32", "64", "bf16x3", "fp32", OP_X3):
        return OP_X3
    raise ValueError(f"unsupported precision {precision!r}: use 32 / 64 (split-bf16 operands, the reference default), 16 (fp16 operands) "
                     "or 'bf16' (bf16 operands)")


class MapperCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("E", "D", "P", "L", "H", "N", "Hm", "W", "use_pos", "op_dtype")]


class Gpt2Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("D", "H", "NL", "V", "Vp", "NPOS", "op_dtype")]


class Gpt2Shape(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "L", "T", "cap", "mode")] + [(n, C.c_float) for n in ("p_embd", "p_attn", "p_resid")] + \
               [("drop_seed", C.c_uint64)]


_P, _I, _L, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_MC, _GC, _GS = C.POINTER(MapperCfg), C.POINTER(Gpt2Cfg), C.POINTER(Gpt2Shape)

# name -> (restype, argtypes): one entry per symbol declared in include/clipcap_hip.h
SIGNATURES = {
    "cc_abi_version": (_I, []),
    "cc_mapper_param_count": (_L, [_MC]),
    "cc_mapper_param_offsets": (_I, [_MC, C.POINTER(_L)]),
    "cc_mapper_ws_bytes": (_L, [_MC, _I, _I]),
    "cc_mapper_sync_weights": (_I, [_MC, _P, _P, _P]),
    "cc_mapper_transpose_weights": (_I, [_MC, _P, _P]),
    "cc_gpt2_sync_weights": (_I, [_GC, _P, _P, _P]),
    "cc_gpt2_transpose_weights": (_I, [_GC, _P, _P]),
    "cc_mapper_fwd": (_I, [_MC, _I, _P, _P, _P, _P, _P, _I, _P]),
    "cc_mapper_attention_probs": (_I, [_MC, _I, _P, _I, _P, _P]),
    "cc_mapper_bwd": (_I, [_MC, _I, _P, _P, _P, _P, _P, _P]),
    "cc_mapper_bwd_range": (_I, [_MC, _I, _P, _P, _P, _P, _P, _I, _I, _P]),
    "cc_gpt2_bwd_range": (_I, [_GC, _GS, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "cc_gpt2_param_count": (_L, [_GC]),
    "cc_gpt2_param_offsets": (_I, [_GC, C.POINTER(_L)]),
    "cc_gpt2_ws_bytes": (_L, [_GC, _GS]),
    "cc_gpt2_embed": (_I, [_GC, _GS, _P, _P, _P, _P, _P]),
    "cc_gpt2_embed_from": (_I, [_GC, _GS, _P, _P, _P, _P]),
    "cc_gpt2_fwd": (_I, [_GC, _GS, _P, _P, _P, _P]),
    "cc_gpt2_logits": (_I, [_GC, _GS, _P, _P, _P, _P, _L, _P]),
    "cc_gpt2_logits_bwd": (_I, [_GC, _GS, _P, _P, _P, _P, _L, _P, _P, _P]),
    "cc_lmhead_ce_fwd": (_I, [_GC, _GS, _P, _P, _P, _P, _P, _P]),
    "cc_lmhead_ce_bwd": (_I, [_GC, _GS, _P, _P, _P, _P, _P, _P, _P]),
    "cc_gpt2_bwd": (_I, [_GC, _GS, _P, _P, _P, _P, _P, _P, _P]),
    "cc_decode_ws_bytes": (_L, [_GC, _I, _I]),
    "cc_decode_fwd": (_I, [_GC, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "cc_decode_part_floats": (_L, [_GC, _I]),
    "cc_decode_fwd_p": (_I, [_GC, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P]),
    "cc_decode_fwd_g": (_I, [_GC, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _L, _P, _P]),
    "cc_beam_step_p": (_I, [_I, _I, _I, _P, _L, _P, _I, _F, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "cc_decode_reorder": (_I, [_GC, _I, _I, _I, _I, _P, _P, _P, _P]),
    "cc_beam_step": (_I, [_I, _I, _I, _P, _L, _F, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "cc_beam_ws_bytes": (_L, [_I, _I, _I]),
    "cc_embed_tokens": (_I, [_GC, _I, _P, _P, _P, _P]),
    "cc_embed_tokens_bwd": (_I, [_GC, _I, _P, _P, _P, _P]),
    "cc_beam_advance": (_I, [_GC, _I, _I, _P, _P, _P, _I, _I, _P, _P, _I, _I, _P, _P, _P, _P]),
    "cc_adamw_step": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P, _P, _P]),
    "cc_adamw_step_cast": (_I, [_I, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P, _P, _P, _P]),
    "cc_cast_op16": (_I, [_I, _P, _P, _L, _P]),
    "cc_grad_wire_pack": (_I, [_P, _P, _L, _P]),
    "cc_grad_wire_unpack": (_I, [_P, _P, _L, _P]),
    "cc_grad_nonfinite": (_I, [_P, _L, _P, _P]),
    "cc_loss_scale_update": (_I, [_P, _P, _F, _F, _I, _P]),
    "cc_dropout_mask": (_I, [C.c_uint64, _I, _I, _F, _L, _P, _P]),
    "cc_sample_step": (_I, [_P, _I, _I, _I, _F, _I, _F, _I, _P, _I, _I, _F, _P, _P, _P, _P]),
    "cc_sample_step_lp": (_I, [_P, _I, _I, _I, _F, _I, _F, _I, _P, _I, _I, _F, _I, _F, _P, _P, _P, _P]),
    "cc_wgrad_scratch_bytes": (_L, []),
    "cc_gemm_wgrad": (_I, [_I, _P, _I, _P, _I, _I, _I, _I, _P, _I, _P, _P]),
    "cc_gemm_tile_mode": (_I, [_I]),
    "cc_gemm_skinny_mode": (_I, [_I]),
    "cc_decode_mode": (_I, [_I]),
    "cc_gemm_op16_f32": (_I, [_I, _I, _I, _P, _I, _P, _I, _I, _I, _I, _P, _I, _P, _I, _P]),
    "cc_layernorm_fwd": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "cc_attention_fwd": (_I, [_I, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "cc_attention_bwd": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "cc_comm_unique_id": (_I, [_P]),
    "cc_comm_create": (_I, [C.POINTER(_P), _I, _I, _P]),
    "cc_allreduce_bucket": (_I, [_P, _P, _L, _I, _P]),
    "cc_broadcast_bucket": (_I, [_P, _P, _L, _I, _I, _P]),
    "cc_reduce_bucket": (_I, [_P, _P, _L, _I, _I, _P]),
    "cc_comm_destroy": (_I, [_P]),
    "cc_comm_count": (_I, [_P, C.POINTER(_I)]),
    "cc_prof_start": (_I, [_I, _I]),
    "cc_prof_stop": (_I, [C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(_I)]),
}

# include/clipcap_hip_lab.h: entry points of the LAB library only (CLIPCAP_HIP_LIB=lab); the product library does not export them
LAB_SIGNATURES = {
    "cc_decode_ws_check": (_I, [_GC, _I, _I, _P, _P]),
    "cc_decode_image_bytes": (_L, [_GC]),
    "cc_decode_image": (_I, [_GC, _P, _P, _P]),
    "cc_decode_xt_image_bytes": (_L, [_GC]),
    "cc_decode_xt_image": (_I, [_GC, _P, _P, _P]),
    "cc_decode_fwd_x": (_I, [_GC, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _L, _P, _P]),
    "cc_decode_last_path": (_I, []),
}

ABI_VERSION = 3       # CC_ABI_VERSION of include/clipcap_hip.h this binding was written against

SITES = {"lmhead_fwd": 1, "lmhead_dgrad": 2, "gpt2_fc_fwd": 3, "gpt2_proj2_fwd": 4, "gpt2_fc_dgrad": 5, "mapper_fc1_fwd": 6,
         "mapper_qkv_fwd": 7, "mapper_wgrad_fc2": 8, "all_gemms": 100}

_lib = None
_lock = threading.Lock()


def lib() -> C.CDLL:
    """Loads the HIP library once; raises HipExtensionMissing (never falls back to a CPU path)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise HipExtensionMissing(
                    f"{LIB_PATH} not found: build it with `make -C clipcap_amd/csrc` (hipcc, gfx950). "
                    "clipcap_amd has no CPU fallback.")
            try:
                l = C.CDLL(LIB_PATH)
            except OSError as e:  # pragma: no cover
                raise HipExtensionMissing(f"cannot load {LIB_PATH}: {e}") from e
            sigs = dict(SIGNATURES, **LAB_SIGNATURES) if IS_LAB else SIGNATURES
            for name, (res, args) in sigs.items():
                fn = getattr(l, name)
                fn.restype = res
                fn.argtypes = args
            if l.cc_abi_version() != ABI_VERSION:
                raise HipExtensionMissing(f"libclipcap_hip.so ABI version {l.cc_abi_version()} != {ABI_VERSION} (include/clipcap_hip.h); rebuild")
            _lib = l
    return _lib


def check(rc: int, what: str = "") -> int:
    if rc < 0:
        raise CCError(f"{what or 'clipcap_hip call'} failed: {ERR.get(rc, rc)} ({rc})")
    return rc
