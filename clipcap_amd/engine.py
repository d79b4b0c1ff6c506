"""Device-side engines: flat parameter arenas + workspaces + calls into libclipcap_hip.so.

``MapperEngine`` / ``Gpt2Engine`` own the arenas the C ABI works on (fp32 master ``w32``, bf16 operand copy ``w16``,
fp32 gradient arena ``g32``) and hand out named views with the reference's state-dict names and layouts
(clipcap/model/mapper.py, clipcap/model/attention.py; HF GPT-2 names).  ``ClipCapEngine`` chains them into the
training step of clipcap/model/model.py:94-113 (forward + backward, no autograd graph).

PyTorch is used for device memory and streams only; all arithmetic runs in the HIP library.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch

from clipcap_amd import _lib
from clipcap_amd._lib import OP_BF16, OP_FP16, OP_X3, Gpt2Cfg, Gpt2Shape, MapperCfg, check, op_dtype_of


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: clipcap_amd runs on an MI355X only (tensor is on {t.device}); there is no CPU fallback")


MAPPER_LAYER_NAMES = ["norm1.weight", "norm1.bias", "attn.to_queries.weight", "attn.to_keys_values.weight", "attn.project.weight",
                      "attn.project.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight",
                      "mlp.fc2.bias"]
GPT2_LAYER_NAMES = ["ln_1.weight", "ln_1.bias", "attn.c_attn.weight", "attn.c_attn.bias", "attn.c_proj.weight", "attn.c_proj.bias",
                    "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias"]


class _Arena:
    """fp32 master + bf16 operand copy (+ lazily a gradient arena and AdamW state) for one parameter family."""

    def __init__(self, n: int, device, sync_fn=None, op_dtype: int = OP_BF16):
        self.n = n
        self.device = torch.device(device)
        self.sync_fn = sync_fn   # (w32_ptr or None, w16_ptr, stream) -> rc : cast + transposed GEMM-weight copies (None: transposed copies only)
        self.op_dtype = op_dtype
        self.w32 = torch.zeros(n, dtype=torch.float32, device=self.device)
        # [0,n): 16-bit cast of the master (bf16 or fp16, cfg.op_dtype); [n,2n): transposed copies of the GEMM weights
        # (include/clipcap_hip.h Conventions).  bf16x3: 6n elements — [hi | lo | hi] row images of every GEMM weight and of its transpose
        self.w16 = self._alloc_w16() if self.device.type == "cuda" else None
        self.g32: Optional[torch.Tensor] = None
        self.m: Optional[torch.Tensor] = None
        self.v: Optional[torch.Tensor] = None
        self._w16_version = -1
        # optimizer-state sharding (train/ddp.py ZeroShard): (rank, [(lo, hi)] per rank, gather(t, ranges)).  m / v then cover the own range only
        self.zero = None
        # tensors aliasing w32 whose in-place writes do NOT bump w32._version: nn.Parameters re-pointed with ``p.data = view``
        # (ArenaModule._rebind) carry their own version counters, so load_state_dict / torch optimizers writing through them
        # would otherwise leave the bf16 operand copy stale.  The dirty stamp is the sum over w32 and every watched alias.
        self.watch: List[torch.Tensor] = []

    def _alloc_w16(self) -> torch.Tensor:
        return torch.zeros((6 if self.op_dtype == OP_X3 else 2) * self.n, dtype=torch.float16 if self.op_dtype == OP_FP16 else torch.bfloat16,
                           device=self.device)

    def mark_dirty(self) -> None:
        """The fp32 master was written behind torch's back (a raw-pointer writer: an RCCL broadcast into the arena, a C-ABI caller):
        the operand copy is rebuilt before its next use."""
        self._w16_version = -1

    def _stamp(self) -> int:
        s = self.w32._version
        for t in self.watch:
            s += t._version
        return s

    def sync_bf16(self):
        """Refresh the bf16 copy if the master changed (through w32, one of its views, or a watched parameter alias)."""
        if self._stamp() != self._w16_version:
            self.refresh_bf16()

    def refresh_bf16(self):
        check(self.sync_fn(_p(self.w32), _p(self.w16), _stream(self.device)), "cc_*_sync_weights")
        self._w16_version = self._stamp()

    def set_op_dtype(self, op_dtype: int) -> None:
        """Switch the operand copy between bf16, fp16 and the split-bf16 images (rebuilt lazily from the fp32 master)."""
        if op_dtype != self.op_dtype:
            self.op_dtype = op_dtype
            if self.w16 is not None:
                self.w16 = self._alloc_w16()
            self._w16_version = -1

    def moved_to(self, device, sync_fn) -> "_Arena":
        """A copy of this arena on `device` carrying the master, the gradient arena and the AdamW moments (resume() followed by
        .to(device) must not lose the optimizer state)."""
        new = _Arena(self.n, device, sync_fn, self.op_dtype)
        new.w32.copy_(self.w32)
        for name in ("g32", "m", "v"):
            t = getattr(self, name)
            if t is not None:
                setattr(new, name, t.to(new.device))
        return new

    def shard_optimizer_state(self, rank: int, world: int, gather) -> None:
        """ZeRO stage 1 (the reference's --deepspeed-strategy, clipcap/train/args.py:87-92, which Lightning hands to DeepSpeed): this rank
        keeps the AdamW moments of elements [lo, hi) of the arena only and steps only those; ``gather(w32, ranges)`` then broadcasts every
        owner's updated slice (ddp.ZeroShard).  Gradients stay all-reduced over the whole arena, so the update of an element is the one
        the unsharded step computes, bit for bit.  Full moments already present (a resumed run) are cut down to the own range."""
        per = (((self.n + world - 1) // world) + 7) // 8 * 8
        ranges = [(min(self.n, r * per), min(self.n, (r + 1) * per)) for r in range(world)]
        lo, hi = ranges[rank]
        for name in ("m", "v"):
            t = getattr(self, name)
            if t is not None and t.numel() == self.n:
                setattr(self, name, t[lo:hi].clone())
        self.zero = (rank, ranges, gather)

    def full_moments(self):
        """(m, v) over the whole arena.  Replicated state: the device tensors themselves.  Sharded state (ZeRO stage 1): a COLLECTIVE (every
        rank calls it) that moves one owner's slice at a time through a slice-sized device buffer into HOST tensors — peak device memory is
        one slice, not two more full-size arenas on every rank (which is what sharding was meant to save)."""
        if self.zero is None or self.m is None:
            return self.m, self.v
        rank, ranges, gather = self.zero
        lo, hi = ranges[rank]
        out = []
        for t in (self.m, self.v):
            full = torch.empty(self.n, dtype=torch.float32)                  # host
            for r, (a, b) in enumerate(ranges):
                if b <= a:
                    continue
                piece = t if r == rank else torch.empty(b - a, dtype=torch.float32, device=self.device)
                gather(piece, [(0, b - a) if q == r else (0, 0) for q in range(len(ranges))])     # one broadcast: owner r's slice
                full[a:b].copy_(piece)
                del piece
            out.append(full)
        return out[0], out[1]

    def grads(self) -> torch.Tensor:
        if self.g32 is None:
            self.g32 = torch.zeros(self.n, dtype=torch.float32, device=self.device)
        return self.g32

    def adamw_step(self, lr: float, step: int, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, grad_scale: float = 1.0,
                   scaler: Optional["LossScaler"] = None):
        """torch.optim.AdamW math (reference model.py:73-77) over the whole arena, refreshing the 16-bit operand copy.
        ``scaler`` (fp16 operands): the gradients carry its loss scale, and a step whose backward overflowed is skipped on the device."""
        if self.zero is not None:                       # sharded optimizer state: own slice, then the owners' slices travel
            rank, ranges, gather = self.zero
            lo, hi = ranges[rank]
            if self.m is None:
                self.m = torch.zeros(hi - lo, dtype=torch.float32, device=self.device)
                self.v = torch.zeros(hi - lo, dtype=torch.float32, device=self.device)
            assert self.m.numel() == hi - lo
            if hi > lo:
                check(_lib.lib().cc_adamw_step(_p(self.w32[lo:hi]), _p(self.grads()[lo:hi]), _p(self.m), _p(self.v), hi - lo, lr, betas[0], betas[1],
                                               eps, weight_decay, 0 if scaler is not None else step, grad_scale,
                                               _p(scaler.scale) if scaler is not None else None,
                                               _p(scaler.found_inf) if scaler is not None else None, _stream(self.device)), "cc_adamw_step")
            gather(self.w32, ranges)
            self.refresh_bf16()
            return
        if self.m is None:
            self.m = torch.zeros_like(self.w32)
            self.v = torch.zeros_like(self.w32)
        # The step also stores the 16-bit cast of every updated parameter (cc_adamw_step_cast), so the operand arena only needs its
        # transposed half rebuilt.  One corner keeps the full refresh: a step that may be SKIPPED on the device (fp16 overflow) leaves
        # the cast untouched, which is only right if the cast was current before the step.
        if os.environ.get("CC_ADAMW_TWO_PASS") or self.op_dtype == OP_X3:   # A/B switch: the separate-cast form; bf16x3: operand images are per matrix
            check(_lib.lib().cc_adamw_step(_p(self.w32), _p(self.grads()), _p(self.m), _p(self.v), self.n, lr, betas[0], betas[1], eps,
                                           weight_decay, 0 if scaler is not None else step, grad_scale, _p(scaler.scale) if scaler is not None else None,
                                           _p(scaler.found_inf) if scaler is not None else None, _stream(self.device)), "cc_adamw_step")
            self.refresh_bf16()
            return
        current = self._stamp() == self._w16_version
        # with a loss scaler the Adam step number is the scaler's device-side count of applied steps (step = 0 asks the kernel for it)
        check(_lib.lib().cc_adamw_step_cast(self.op_dtype, _p(self.w32), _p(self.grads()), _p(self.m), _p(self.v), self.n, lr, betas[0],
                                            betas[1], eps, weight_decay, 0 if scaler is not None else step, grad_scale, _p(scaler.scale) if scaler is not None else None,
                                            _p(scaler.found_inf) if scaler is not None else None, _p(self.w16), _stream(self.device)),
              "cc_adamw_step_cast")
        if scaler is None or current:
            check(self.sync_fn(None, _p(self.w16), _stream(self.device)), "cc_*_transpose_weights")
            self._w16_version = self._stamp()
        else:
            self.refresh_bf16()


class LossScaler:
    """Dynamic loss scale for fp16-operand training, torch.cuda.amp.GradScaler semantics (init 2^16, x2 every 2000 good steps, x0.5
    on overflow) — what Lightning wraps around the reference's model for ``--fp-precision 16`` — kept entirely on the device:
    ``state`` = [scale, good-step counter, applied-step counter], ``found_inf`` is raised by cc_grad_nonfinite and read by cc_adamw_step (which then skips
    the step) and by cc_loss_scale_update.  No host synchronisation anywhere."""

    def __init__(self, device, init_scale: float = 65536.0, growth: float = 2.0, backoff: float = 0.5, interval: int = 2000):
        self.device = torch.device(device)
        # [scale, good steps since the last change, optimizer steps actually APPLIED (skipped ones do not count: Adam's bias correction
        # follows this counter, as torch.cuda.amp.GradScaler + torch.optim.AdamW do)]
        self.state = torch.tensor([init_scale, 0.0, 0.0], dtype=torch.float32, device=self.device)
        self.found_inf = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.growth, self.backoff, self.interval = growth, backoff, interval

    @property
    def scale(self) -> torch.Tensor:
        return self.state[0:1]

    def check(self, arena: "_Arena") -> None:
        """Raise found_inf if the arena's (scaled) gradients hold an inf / nan.  Multi-GPU: call after the all-reduce."""
        check(_lib.lib().cc_grad_nonfinite(_p(arena.grads()), arena.n, _p(self.found_inf), _stream(self.device)), "cc_grad_nonfinite")

    def update(self) -> None:
        """After the optimizer steps of this iteration: adjust the scale and clear found_inf."""
        check(_lib.lib().cc_loss_scale_update(_p(self.state), _p(self.found_inf), self.growth, self.backoff, self.interval,
                                              _stream(self.device)), "cc_loss_scale_update")


class MapperEngine:
    """TransformerMapper / TransformerMapperWindowed (reference clipcap/model/mapper.py:113-160) on the HIP library."""

    def __init__(self, E: int, D: int, prefix_length: int, projection_length: int, num_heads: int, num_layers: int, window: int = 1,
                 use_pos: bool = False, device="cpu", precision=None):
        self.dims = dict(E=E, D=D, P=projection_length, L=prefix_length, H=num_heads, N=num_layers, Hm=int(D * 2.0), W=window,
                         use_pos=int(bool(use_pos) and window > 1))
        self.op_dtype = op_dtype_of(precision)
        self.cfg = MapperCfg(op_dtype=self.op_dtype, **self.dims)
        l = _lib.lib()
        n = l.cc_mapper_param_count(C.byref(self.cfg))
        if n < 0:
            raise _lib.CCError(f"unsupported mapper configuration {self.dims}: dims must be multiples of 8 (head dim too)")
        self.arena = _Arena(n, device, self._sync, self.op_dtype)
        offs = (C.c_int64 * (4 + 12 * num_layers))()
        check(l.cc_mapper_param_offsets(C.byref(self.cfg), offs))
        self.offsets = list(offs)
        self._ws: Dict[Tuple[int, int], torch.Tensor] = {}

    def _sync(self, w32, w16, st):
        if w32 is None:          # the cast half is current (written by the optimizer step): transposed copies only
            return _lib.lib().cc_mapper_transpose_weights(C.byref(self.cfg), w16, st)
        return _lib.lib().cc_mapper_sync_weights(C.byref(self.cfg), w32, w16, st)

    # ---- named views (reference state-dict names / layouts) ----
    def shapes(self) -> List[Tuple[str, int, Tuple[int, ...]]]:
        d = self.dims
        E, D, P, L, N, Hm, W = d["E"], d["D"], d["P"], d["L"], d["N"], d["Hm"], d["W"]
        out = [("linear.weight", self.offsets[0], (P * D, E)), ("linear.bias", self.offsets[1], (P * D,)),
               ("prefix_const", self.offsets[2], (L, D))]
        if self.offsets[3] >= 0:
            out.append(("pos_embeddings", self.offsets[3], (W * P, D)))
        shp = [(D,), (D,), (D, D), (2 * D, D), (D, D), (D,), (D,), (D,), (Hm, D), (Hm,), (D, Hm), (D,)]
        for i in range(N):
            for j, nm in enumerate(MAPPER_LAYER_NAMES):
                out.append((f"transformer.layers.{i}.{nm}", self.offsets[4 + 12 * i + j], shp[j]))
        return out

    def views(self, arena: torch.Tensor) -> Dict[str, torch.Tensor]:
        res = {}
        for name, off, shape in self.shapes():
            n = 1
            for s in shape:
                n *= s
            res[name] = arena[off:off + n].view(shape)
        return res

    def to(self, device):
        device = torch.device(device)
        if device == self.arena.device:
            return self
        self.arena = self.arena.moved_to(device, self._sync)
        self._ws.clear()
        return self

    def set_precision(self, precision) -> None:
        """16 -> fp16 operands (the reference's --fp-precision 16), 32 / 64 -> split-bf16 operands (the reference's default precision),
        'bf16' -> bf16 operands (_lib.op_dtype_of).  fp32 master unchanged."""
        self.op_dtype = op_dtype_of(precision)
        self.cfg.op_dtype = self.op_dtype
        self.arena.set_op_dtype(self.op_dtype)
        self._ws.clear()

    def _workspace(self, B: int, save: int) -> torch.Tensor:
        key = (B, save)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = _lib.lib().cc_mapper_ws_bytes(C.byref(self.cfg), B, save)
            check(nbytes, "cc_mapper_ws_bytes")
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.arena.device)
            self._ws[key] = ws
        return ws

    def forward(self, emb: torch.Tensor, save: bool = False) -> torch.Tensor:
        """emb fp32 (B,E) or (B,W,E) -> prefix fp32 (B,L,D)."""
        _require_cuda(self.arena.w32, "MapperEngine.forward")
        d = self.dims
        emb = emb.to(device=self.arena.device, dtype=torch.float32).contiguous()
        B = emb.shape[0]
        if emb.numel() != B * d["W"] * d["E"]:
            raise ValueError(f"expected embeddings of shape ({B},{d['W']},{d['E']}), got {tuple(emb.shape)}")
        self.arena.sync_bf16()
        out = torch.empty(B, d["L"], d["D"], dtype=torch.float32, device=self.arena.device)
        ws = self._workspace(B, int(save))
        check(_lib.lib().cc_mapper_fwd(C.byref(self.cfg), B, _p(self.arena.w32), _p(self.arena.w16), _p(emb), _p(ws), _p(out), int(save),
                                      _stream(self.arena.device)), "cc_mapper_fwd")
        self._last = (B, emb)  # keep the input alive until backward has consumed the workspace
        if save:
            self._saved_batch = B
            self._fwd_serial = getattr(self, "_fwd_serial", 0) + 1
        return out

    def attention_probs(self, B: int) -> List[torch.Tensor]:
        """Per layer, the attention probabilities (B, S, S, H) of the last forward(save=True) with batch B — the second return value
        of the reference's MultiHeadAttention.forward (attention.py:32-42)."""
        ws = self._ws.get((B, 1))
        if ws is None or getattr(self, "_saved_batch", None) != B:
            raise RuntimeError("MapperEngine.attention_probs needs a preceding forward(save=True) with the same batch")
        d = self.dims
        S = d["W"] * d["P"] + d["L"]
        outs = []
        for l in range(d["N"]):
            out = torch.empty(B, S, S, d["H"], dtype=torch.float32, device=self.arena.device)
            check(_lib.lib().cc_mapper_attention_probs(C.byref(self.cfg), B, _p(ws), l, _p(out), _stream(self.arena.device)),
                  "cc_mapper_attention_probs")
            outs.append(out)
        return outs

    def layer_span(self, l: int) -> Tuple[int, int]:
        """[lo, hi) element range of layer l's parameters in the arenas (layers are contiguous, head tensors come first)."""
        lo = self.offsets[4 + 12 * l]
        hi = self.offsets[4 + 12 * (l + 1)] if l + 1 < self.dims["N"] else self.arena.n
        return lo, hi

    def backward(self, dout: torch.Tensor, on_layers_done=None, group: int = 2):
        """Accumulates d loss / d params into the gradient arena; needs a preceding forward(save=True).
        on_layers_done(lo, hi): called right after the kernels of a slice of layers are enqueued, with the arena element range
        whose gradients are then final — the hook the DDP reducer uses to overlap its all-reduce with the rest of backward."""
        B = dout.shape[0]
        dout = dout.to(dtype=torch.float32).contiguous()
        ws = self._ws.get((B, 1))
        if ws is None:
            raise RuntimeError("MapperEngine.backward without forward(save=True)")
        l = _lib.lib()
        a = self.arena
        N = self.dims["N"]
        if on_layers_done is None:
            check(l.cc_mapper_bwd(C.byref(self.cfg), B, _p(a.w32), _p(a.w16), _p(ws), _p(dout), _p(a.grads()), _stream(a.device)), "cc_mapper_bwd")
            return
        hi = N
        while hi > 0:
            lo = max(0, hi - group)
            check(l.cc_mapper_bwd_range(C.byref(self.cfg), B, _p(a.w32), _p(a.w16), _p(ws), _p(dout), _p(a.grads()), hi, lo, _stream(a.device)),
                  "cc_mapper_bwd_range")
            if lo > 0:
                on_layers_done(self.layer_span(lo)[0], self.layer_span(hi - 1)[1])
            else:   # the l_lo == 0 call also produced the head tensors' gradients
                on_layers_done(0, self.layer_span(hi - 1)[1])
            hi = lo
        if N == 0:
            check(l.cc_mapper_bwd_range(C.byref(self.cfg), B, _p(a.w32), _p(a.w16), _p(ws), _p(dout), _p(a.grads()), 0, 0, _stream(a.device)),
                  "cc_mapper_bwd_range")
            on_layers_done(0, a.n)


class Gpt2Engine:
    """HF GPT2LMHeadModel arithmetic (transformers modeling_gpt2.py) on the HIP library; arena keeps HF layouts."""

    def __init__(self, n_embd: int, n_head: int, n_layer: int, vocab_size: int, n_positions: int, device="cpu", precision=None):
        self.dims = dict(D=n_embd, H=n_head, NL=n_layer, V=vocab_size, Vp=(vocab_size + 127) // 128 * 128, NPOS=n_positions)
        self.op_dtype = op_dtype_of(precision)
        self.cfg = Gpt2Cfg(op_dtype=self.op_dtype, **self.dims)
        l = _lib.lib()
        n = l.cc_gpt2_param_count(C.byref(self.cfg))
        if n < 0:
            raise _lib.CCError(f"unsupported GPT-2 configuration {self.dims}")
        self.arena = _Arena(n, device, self._sync, self.op_dtype)
        offs = (C.c_int64 * (2 + 12 * n_layer + 2))()
        check(l.cc_gpt2_param_offsets(C.byref(self.cfg), offs))
        self.offsets = list(offs)
        self._ws: Dict[int, torch.Tensor] = {}

    def _sync(self, w32, w16, st):
        if w32 is None:
            return _lib.lib().cc_gpt2_transpose_weights(C.byref(self.cfg), w16, st)
        return _lib.lib().cc_gpt2_sync_weights(C.byref(self.cfg), w32, w16, st)

    def decode_images(self) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
        """LAB LIBRARY ONLY (CLIPCAP_HIP_LIB=lab; (None, None) with the product library).  (wimg, wteam) for cc_decode_fwd_x
        (include/clipcap_hip_lab.h), rebuilt whenever the operand copy changes:
        wimg  — the block GEMM weights in MFMA fragment order (cc_decode_image; cc_decode_mode bit 3): the decode GEMMs whose K is split over
                the waves load the weight operand global -> VGPR from it;
        wteam — the XCD-team engine's image (cc_decode_xt_image; lab build, cc_decode_mode bit 2).
        None where the library does not cover this model (width, operand type), the switch is off, or the build lacks the engine."""
        a = self.arena
        if not _lib.IS_LAB:
            return None, None
        l = _lib.lib()
        mode = l.cc_decode_mode(-1)
        if a.device.type != "cuda" or not (mode & 12):
            return None, None
        a.sync_bf16()
        key = (a._w16_version, a.op_dtype, a.w16.data_ptr(), mode & 12)
        if getattr(self, "_img_key", None) != key:
            out = []
            for bit, nbytes_fn, build_fn, name in ((8, l.cc_decode_image_bytes, l.cc_decode_image, "cc_decode_image"),
                                                  (4, l.cc_decode_xt_image_bytes, l.cc_decode_xt_image, "cc_decode_xt_image")):
                nbytes = nbytes_fn(C.byref(self.cfg)) if mode & bit else 0
                if nbytes <= 0:
                    out.append(None)
                    continue
                img = torch.empty(nbytes // 2, dtype=a.w16.dtype, device=a.device)
                check(build_fn(C.byref(self.cfg), _p(a.w16), _p(img), _stream(a.device)), name)
                out.append(img)
            self._imgs = tuple(out)
            self._img_key = key
        return self._imgs

    def shapes(self) -> List[Tuple[str, int, Tuple[int, ...]]]:
        d = self.dims
        D, NL, V, NPOS = d["D"], d["NL"], d["V"], d["NPOS"]
        out = [("transformer.wte.weight", self.offsets[0], (V, D)), ("transformer.wpe.weight", self.offsets[1], (NPOS, D))]
        shp = [(D,), (D,), (D, 3 * D), (3 * D,), (D, D), (D,), (D,), (D,), (D, 4 * D), (4 * D,), (4 * D, D), (D,)]
        for i in range(NL):
            for j, nm in enumerate(GPT2_LAYER_NAMES):
                out.append((f"transformer.h.{i}.{nm}", self.offsets[2 + 12 * i + j], shp[j]))
        out.append(("transformer.ln_f.weight", self.offsets[2 + 12 * NL], (D,)))
        out.append(("transformer.ln_f.bias", self.offsets[3 + 12 * NL], (D,)))
        return out

    views = MapperEngine.views
    to = MapperEngine.to
    set_precision = MapperEngine.set_precision

    def layer_span(self, l: int) -> Tuple[int, int]:
        lo = self.offsets[2 + 12 * l]
        hi = self.offsets[2 + 12 * (l + 1)]      # the entry after the last layer is ln_f.weight
        return lo, hi

    def shape(self, B: int, L: int, T: int, cap: int, mode: int, dropout=None) -> Gpt2Shape:
        """dropout: optional (p_embd, p_attn, p_resid, seed) of this pass (full finetune in train mode); None = eval behaviour."""
        if dropout is None:
            return Gpt2Shape(B, L, T, cap, mode, 0.0, 0.0, 0.0, 0)
        return Gpt2Shape(B, L, T, cap, mode, float(dropout[0]), float(dropout[1]), float(dropout[2]), int(dropout[3]) & 0xFFFFFFFFFFFFFFFF)

    def workspace(self, shp: Gpt2Shape) -> torch.Tensor:
        """One buffer per mode, grown on demand: the library carves its layout from the base pointer on every call, so a buffer
        sized for the largest shape seen serves all smaller ones (caption length varies from batch to batch in real training)."""
        nbytes = _lib.lib().cc_gpt2_ws_bytes(C.byref(self.cfg), C.byref(shp))
        check(nbytes, "cc_gpt2_ws_bytes")
        ws = self._ws.get(shp.mode)
        if ws is None or ws.numel() < nbytes or ws.device != self.arena.device:
            self._ws.pop(shp.mode, None)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.arena.device)
            self._ws[shp.mode] = ws
        # every hand-out of a mode's buffer may overwrite the activations a differentiable .logits pass left in it: the autograd
        # bridge compares this counter (logits(save_mode) / logits_backward)
        self._ws_uses = getattr(self, "_ws_uses", 0) + 1
        return ws

    # -- inference-style forward: logits for all rows (ClipCapModel.forward / language_model(inputs_embeds=...)) --
    def logits(self, inputs_embeds: torch.Tensor, save_mode: int = 0) -> torch.Tensor:
        """inputs_embeds fp32 (B,T,D) (no positional embedding) -> fp32 logits (B,T,V).  save_mode 1 / 2 keeps the activations for
        logits_backward (1: gradient wrt inputs_embeds only, 2: + every GPT-2 weight gradient)."""
        _require_cuda(self.arena.w32, "Gpt2Engine.logits")
        l = _lib.lib()
        x = inputs_embeds.to(device=self.arena.device, dtype=torch.float32).contiguous()
        B, T, D = x.shape
        shp = self.shape(B, T, T, 0, 0) if not save_mode else self.shape(B, 0, T, T, save_mode)
        if save_mode:
            self._logits_pass = (shp, getattr(self, "_logits_pass", (None, 0, 0))[1] + 1, 0)
        ws = self.workspace(shp)
        self.arena.sync_bf16()
        st = _stream(self.arena.device)
        a = self.arena
        check(l.cc_gpt2_embed_from(C.byref(self.cfg), C.byref(shp), _p(a.w32), _p(x), _p(ws), st), "cc_gpt2_embed_from")
        check(l.cc_gpt2_fwd(C.byref(self.cfg), C.byref(shp), _p(a.w32), _p(a.w16), _p(ws), st), "cc_gpt2_fwd")
        Vp = self.dims["Vp"]
        out = torch.empty(B * T, Vp, dtype=torch.float32, device=a.device)
        check(l.cc_gpt2_logits(C.byref(self.cfg), C.byref(shp), _p(a.w32), _p(a.w16), _p(ws), _p(out), Vp, st), "cc_gpt2_logits")
        if save_mode:
            self._logits_pass = (shp, self._logits_pass[1], self._ws_uses)
        return out.view(B, T, Vp)[:, :, : self.dims["V"]]

    def logits_backward(self, dlogits: torch.Tensor) -> torch.Tensor:
        """d loss / d inputs_embeds (B,T,D) for the last logits(..., save_mode >= 1) pass; save_mode 2 also ACCUMULATES the GPT-2 weight
        gradients into the arena's g32 (cc_gpt2_logits_bwd)."""
        shp, _, uses = self._logits_pass
        if uses != self._ws_uses:
            raise RuntimeError("Gpt2Engine.logits_backward: another pass has used the activation workspace since the forward; "
                               "call backward() before running the language model again")
        a = self.arena
        dl = dlogits.to(device=a.device, dtype=torch.float32).contiguous()
        B, T, V = dl.shape
        assert (B, T) == (shp.B, shp.T) and V == self.dims["V"]
        dx0 = torch.empty(B, T, self.dims["D"], dtype=torch.float32, device=a.device)
        ws = self.workspace(shp)
        scale = keep = None
        if self.op_dtype == OP_FP16:
            # fp16 operands: d logits of a mean cross-entropy are p / N ~ 1e-9 .. 1e-5, below fp16's normal range (6e-5) — the whole
            # backward runs under a power-of-two scale chosen on the device from max |d logits| (no host sync) and divided out of dx0
            # and of the weight gradients afterwards, as the fused trainer does with its LossScaler
            amax = dl.abs().amax().clamp_min(1e-30)
            scale = torch.exp2(torch.floor(torch.log2(256.0 / amax)))
            dl = dl * scale
            if shp.mode == 2:
                keep = a.grads().clone()
                a.grads().zero_()
        check(_lib.lib().cc_gpt2_logits_bwd(C.byref(self.cfg), C.byref(shp), _p(a.w32), _p(a.w16), _p(ws), _p(dl), V, _p(dx0),
                                            _p(a.grads()) if shp.mode == 2 else None, _stream(a.device)), "cc_gpt2_logits_bwd")
        if scale is not None:
            dx0 = dx0 / scale
            if keep is not None:
                a.grads().div_(scale).add_(keep)
        return dx0


class ClipCapEngine:
    """Training step of ClipCapModel (reference model.py:94-113) as straight-line forward + backward kernel chains."""

    def __init__(self, mapper: MapperEngine, gpt2: Gpt2Engine, train_lm: bool):
        self.mapper = mapper
        self.gpt2 = gpt2
        self.train_lm = train_lm
        self.stats: Optional[torch.Tensor] = None
        # fp16 operands: the backward pass runs under a dynamic loss scale (gradients in g32 are scale x the true ones until
        # optimizer_step divides it out); bf16 operands need none
        self.scaler: Optional[LossScaler] = None

    def _scaler(self, dev) -> Optional[LossScaler]:
        if OP_FP16 not in (self.gpt2.op_dtype, self.mapper.op_dtype):
            return None
        if self.scaler is None or self.scaler.device != dev:
            self.scaler = LossScaler(dev)
        return self.scaler

    def arenas(self) -> List[_Arena]:
        return [self.mapper.arena] + ([self.gpt2.arena] if self.train_lm else [])

    def optimizer_step(self, lr: float, step: int, sync_flag=None, **adamw_kw) -> None:
        """AdamW over every trained arena (reference model.py:67-91).  With fp16 operands: overflow check of the scaled gradients
        (after any all-reduce), the step is skipped on the device if they overflowed, then the loss scale is adjusted.
        ``sync_flag(found_inf)``: optional in-place SUM over the ranks (ddp.GradReducer.reduce_flag) — with partitioned gradients
        (ZeRO stage 2) a rank only sees its own slice summed, so the ranks agree on the overflow flag before anyone steps."""
        sc = self.scaler
        if sc is not None:
            for a in self.arenas():
                sc.check(a)
            if sync_flag is not None:
                sync_flag(sc.found_inf)
        for a in self.arenas():
            a.adamw_step(lr, step, scaler=sc, **adamw_kw)
        if sc is not None:
            sc.update()

    def forward_backward(self, tokens: torch.Tensor, embeds: torch.Tensor, reduce_stats=None, backward: bool = True,
                         on_grads_ready=None, dropout=None) -> torch.Tensor:
        """tokens int64 (B,cap) padded with -1; embeds fp32 (B,E)/(B,W,E).

        Returns the mean loss over kept targets (device scalar).  Gradients are ACCUMULATED into the arenas' g32.
        ``reduce_stats(stats)``: optional in-place all-reduce of the 2-float [loss_sum, kept_count] tensor, so the divisor is the
        global kept-token count (N-rank == 1-rank gradients).
        ``on_grads_ready(arena_index, lo, hi)``: optional; called as soon as the kernels producing gradient elements [lo, hi) of
        arena 0 (mapper) / 1 (GPT-2) are enqueued, in the order backward finishes them (GPT-2 top layers first, mapper last), so
        the caller can launch the all-reduce of that slice underneath the remaining backward kernels.
        ``dropout``: optional (p_embd, p_attn, p_resid, seed) — GPT-2 train-mode dropout for this step (full finetune; the masks
        are a hash of (seed, site, layer, element) regenerated by the backward kernels; the setting travels in this pass's
        cc_gpt2_shape, so concurrent engines on other streams are unaffected).
        With fp16 operands the gradients left in the arenas are multiplied by the current loss scale (self.scaler.scale);
        optimizer_step() divides it out.
        """
        l = _lib.lib()
        g, m = self.gpt2, self.mapper
        dev = g.arena.device
        _require_cuda(g.arena.w32, "ClipCapEngine.forward_backward")
        tokens = tokens.to(device=dev, dtype=torch.int64).contiguous()
        B, cap = tokens.shape
        L = m.dims["L"]
        T = L + cap
        mode = 2 if self.train_lm else 1
        shp = g.shape(B, L, T, cap, mode, dropout)
        ws = g.workspace(shp)
        g.arena.sync_bf16()
        st = _stream(dev)
        ga = g.arena
        prefix = m.forward(embeds, save=True)
        return self._forward_backward(l, g, m, ga, shp, ws, st, prefix, tokens, reduce_stats, backward, on_grads_ready, dev)

    def _forward_backward(self, l, g, m, ga, shp, ws, st, prefix, tokens, reduce_stats, backward, on_grads_ready, dev):
        check(l.cc_gpt2_embed(C.byref(g.cfg), C.byref(shp), _p(ga.w32), _p(prefix), _p(tokens), _p(ws), st), "cc_gpt2_embed")
        check(l.cc_gpt2_fwd(C.byref(g.cfg), C.byref(shp), _p(ga.w32), _p(ga.w16), _p(ws), st), "cc_gpt2_fwd")
        if self.stats is None or self.stats.device != dev:
            self.stats = torch.zeros(2, dtype=torch.float32, device=dev)
        stats = self.stats
        check(l.cc_lmhead_ce_fwd(C.byref(g.cfg), C.byref(shp), _p(ga.w32), _p(ga.w16), _p(ws), _p(tokens), _p(stats), st), "cc_lmhead_ce_fwd")
        if reduce_stats is not None:
            reduce_stats(stats)
        if backward:
            g32 = ga.grads() if self.train_lm else None
            denom = stats[1:2]
            sc = self._scaler(dev)
            check(l.cc_lmhead_ce_bwd(C.byref(g.cfg), C.byref(shp), _p(ga.w32), _p(ga.w16), _p(ws), _p(denom),
                                     _p(sc.scale) if sc is not None else None, _p(g32), st), "cc_lmhead_ce_bwd")
            dprefix = torch.empty_like(prefix)
            if on_grads_ready is None or not self.train_lm:
                check(l.cc_gpt2_bwd(C.byref(g.cfg), C.byref(shp), _p(ga.w32), _p(ga.w16), _p(ws), _p(tokens), _p(dprefix), _p(g32), st), "cc_gpt2_bwd")
            else:
                NL, grp = g.dims["NL"], 2
                on_grads_ready(1, g.layer_span(NL - 1)[1], ga.n)                    # ln_f (from cc_lmhead_ce_bwd)
                hi = NL
                while hi > 0:
                    lo = max(0, hi - grp)
                    check(l.cc_gpt2_bwd_range(C.byref(g.cfg), C.byref(shp), _p(ga.w32), _p(ga.w16), _p(ws), _p(tokens), _p(dprefix), _p(g32),
                                              hi, lo, st), "cc_gpt2_bwd_range")
                    on_grads_ready(1, g.layer_span(lo)[0] if lo > 0 else 0, g.layer_span(hi - 1)[1])   # lo == 0: + wte / wpe
                    hi = lo
            m.backward(dprefix, on_layers_done=(None if on_grads_ready is None else (lambda lo, hi: on_grads_ready(0, lo, hi))))
        return stats[0] / stats[1].clamp_min(1.0)

    def zero_grad(self):
        if self.mapper.arena.g32 is not None:
            self.mapper.arena.g32.zero_()
        if self.train_lm and self.gpt2.arena.g32 is not None:
            self.gpt2.arena.g32.zero_()


class DecodeSession:
    """KV-cached GPT-2 decode state for R rows (replaces the per-step full re-forward of the reference's inference/base.py:81).
    The cache is bf16 [n_layer][2][R][ctx_max][D], owned here.  A beam reorder (base.py:93,113) does not move cache data: a small
    int32 ancestry table ``row_map[r][j]`` names the cache row that holds position j of logical row r, and ``reorder`` permutes it."""

    def __init__(self, gpt2: Gpt2Engine, rows: int, ctx_max: int):
        _require_cuda(gpt2.arena.w32, "DecodeSession")
        self.g = gpt2
        self.R = rows
        self.ctx_max = min(ctx_max, gpt2.dims["NPOS"])
        self.pos = 0
        d = gpt2.dims
        dev = gpt2.arena.device
        # 16-bit operand type; fp32 in the bf16x3 mode (include/clipcap_hip.h OPERAND TYPE)
        self.kv = torch.empty(d["NL"] * 2 * rows * self.ctx_max * d["D"], dtype=torch.float32 if gpt2.op_dtype == OP_X3 else gpt2.arena.w16.dtype,
                              device=dev)
        self.row_map = torch.arange(rows, dtype=torch.int32, device=dev).view(rows, 1).repeat(1, self.ctx_max).contiguous()
        self._ws: Dict[int, torch.Tensor] = {}
        self._logits: Optional[torch.Tensor] = None

    def _workspace(self, tn: int) -> torch.Tensor:
        ws = self._ws.get(tn)
        if ws is None:
            nbytes = _lib.lib().cc_decode_ws_bytes(C.byref(self.g.cfg), self.R, tn)
            check(nbytes, "cc_decode_ws_bytes")
            ws = torch.zeros(nbytes, dtype=torch.uint8, device=self.g.arena.device)     # zeroed: cc_decode_ws_check's error word starts clear
            self._ws[tn] = ws
        return ws

    def forward(self, x: torch.Tensor, rows: Optional[int] = None, partials: bool = False, group: int = 1) -> torch.Tensor:
        """x fp32 (R', Tnew, D) input embeddings (no positional term), R' <= R active rows -> fp32 logits of the last new
        position (R', V).  Valid until the next forward().  ``partials``: also keep the lm_head epilogue's per-64-column softmax
        partials of these logits in ``self.lpart`` = (tensor, npart) for beam_step (cc_decode_fwd_g / cc_beam_step_p).
        ``group``: consecutive rows that share ancestry (beam search: the beam width) — a performance hint, results do not depend on it."""
        g = self.g
        x = x.to(device=g.arena.device, dtype=torch.float32).contiguous()
        Ra, tn, D = x.shape
        assert Ra <= self.R and D == g.dims["D"]
        if self.pos + tn > self.ctx_max:
            raise RuntimeError(f"decode context overflow: {self.pos}+{tn} > {self.ctx_max}")
        g.arena.sync_bf16()
        Vp = g.dims["Vp"]
        if self._logits is None:
            self._logits = torch.empty(self.R, Vp, dtype=torch.float32, device=g.arena.device)
        logits = self._logits[:Ra]
        if Ra != self.R:   # fewer active rows (prefill with one row per sample): the cache row stride is still R rows
            raise RuntimeError("partial-row forward is expressed through a narrower session; use expand()")
        self.lpart = None
        if partials:
            if getattr(self, "_lpart", None) is None:
                n = _lib.lib().cc_decode_part_floats(C.byref(g.cfg), self.R)
                check(n, "cc_decode_part_floats")
                self._lpart = torch.empty(n, dtype=torch.float32, device=g.arena.device)
            self.lpart = (self._lpart, self._lpart.numel() // (2 * self.R))
        grp = int(group) if self.R % max(1, int(group)) == 0 else 1
        wimg, wteam = g.decode_images() if (tn == 1 and _lib.IS_LAB) else (None, None)     # lab library: weight images of the decode experiments
        if wimg is not None or wteam is not None:
            check(_lib.lib().cc_decode_fwd_x(C.byref(g.cfg), Ra, tn, self.pos, self.ctx_max, _p(g.arena.w32), _p(g.arena.w16),
                                            _p(wimg) if wimg is not None else None, _p(wteam) if (wteam is not None and grp >= 2) else None, _p(x), _p(self.kv),
                                            _p(self.row_map), grp, _p(self._workspace(tn)), _p(logits), Vp,
                                            _p(self._lpart) if partials else None, _stream(g.arena.device)), "cc_decode_fwd_x")
        else:
            check(_lib.lib().cc_decode_fwd_g(C.byref(g.cfg), Ra, tn, self.pos, self.ctx_max, _p(g.arena.w32), _p(g.arena.w16), _p(x), _p(self.kv),
                                            _p(self.row_map), grp, _p(self._workspace(tn)), _p(logits), Vp,
                                            _p(self._lpart) if partials else None, _stream(g.arena.device)), "cc_decode_fwd_g")
        self.pos += tn
        return logits[:, : g.dims["V"]]

    def check(self, tn: int = 1) -> None:
        """Lab library: synchronises and raises if the last single-position step on this session's workspace gave up on an in-launch
        hand-off of the persistent-launch experiments (cc_decode_ws_check).  The product path has no in-launch hand-offs: nothing to check."""
        if _lib.IS_LAB and tn in self._ws:
            check(_lib.lib().cc_decode_ws_check(C.byref(self.g.cfg), self.R, tn, _p(self._ws[tn]), _stream(self.g.arena.device)), "cc_decode_ws_check")

    def reorder(self, src_rows: torch.Tensor) -> "DecodeSession":
        """Logical row r continues the history of logical row src_rows[r] (same row count): permutes the ancestry table in place."""
        src = src_rows.to(device=self.g.arena.device, dtype=torch.int64)
        if self.pos > 0:
            self.row_map[:, : self.pos] = self.row_map.index_select(0, src)[:, : self.pos]
        return self

    def beam_advance(self, beam: int, next_tok: torch.Tensor, src_local: Optional[torch.Tensor], wte: torch.Tensor, step: int,
                     tokens_in: torch.Tensor, tokens_out: torch.Tensor, x_out: torch.Tensor) -> None:
        """Everything between two beam steps in one launch (cc_beam_advance; reference inference/base.py:104-117): row r continues row
        (r // beam) * beam + src_local[r] (None: itself) — the ancestry table is permuted (into a second buffer, then swapped),
        tokens_out[r] = tokens_in[that row][:step] + [next_tok[r]] (int32 (R, n) buffers) and x_out (R, 1, D) fp32 = wte[next_tok]."""
        if getattr(self, "_row_map_alt", None) is None:
            self._row_map_alt = torch.empty_like(self.row_map)
        check(_lib.lib().cc_beam_advance(C.byref(self.g.cfg), self.R, beam, _p(wte), _p(next_tok), _p(src_local) if src_local is not None else None,
                                        self.pos, self.ctx_max, _p(self.row_map), _p(self._row_map_alt), step, tokens_out.stride(0),
                                        _p(tokens_in), _p(tokens_out), _p(x_out), _stream(self.g.arena.device)), "cc_beam_advance")
        self.row_map, self._row_map_alt = self._row_map_alt, self.row_map

    def expand(self, src_rows: torch.Tensor, rows_out: int) -> "DecodeSession":
        """A wider session (rows_out rows) whose row r starts from this session's row src_rows[r] (beam fan-out after the prefill,
        base.py:93).  The prefix K/V are copied once (cc_decode_reorder)."""
        out = DecodeSession(self.g, rows_out, self.ctx_max)
        out.pos = self.pos
        src = self.row_map[:, 0].index_select(0, src_rows.to(device=self.g.arena.device, dtype=torch.int64)).to(torch.int32).contiguous() \
            if self.pos > 0 else src_rows.to(device=self.g.arena.device, dtype=torch.int32).contiguous()
        check(_lib.lib().cc_decode_reorder(C.byref(self.g.cfg), self.R, rows_out, self.pos, self.ctx_max, _p(self.kv), _p(out.kv), _p(src),
                                          _stream(self.g.arena.device)), "cc_decode_reorder")
        if self.pos > 0:
            # rows fanned out from one source hold identical prefix K / V: their ancestry tables all name the FIRST copy, so that the
            # beam-group attention step (cc_decode_fwd_g) reads one copy per group instead of one per row
            idx = torch.arange(rows_out, dtype=torch.int32, device=src.device)
            first = torch.full((int(self.R),), rows_out, dtype=torch.int32, device=src.device).scatter_reduce(0, src.to(torch.int64), idx, "amin")
            out.row_map[:, : self.pos] = first.index_select(0, src.to(torch.int64)).view(rows_out, 1)
        return out


def embed_tokens(gpt2: "Gpt2Engine", tokens: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out fp32 (R, 1, D) or (R, D) = wte[tokens] for int32 ``tokens`` (R,) in one launch (cc_embed_tokens; reference inference/base.py:117,184)."""
    wte = gpt2.arena.w32[gpt2.offsets[0]:]
    check(_lib.lib().cc_embed_tokens(C.byref(gpt2.cfg), tokens.numel(), _p(wte), _p(tokens), _p(out), _stream(gpt2.arena.device)), "cc_embed_tokens")
    return out


def embed_tokens_bwd(gpt2: "Gpt2Engine", tokens: torch.Tensor, dout: torch.Tensor) -> torch.Tensor:
    """fp32 (V, D) gradient of ``embed_tokens`` with respect to wte: rows of ``dout`` (R, D) scattered by int32 ``tokens`` (R,), rows that
    share an id accumulate (cc_embed_tokens_bwd; fp32 atomics, like torch's embedding backward on a GPU)."""
    D, V, Vp = gpt2.dims["D"], gpt2.dims["V"], gpt2.dims["Vp"]
    dw = torch.zeros(Vp, D, dtype=torch.float32, device=gpt2.arena.device)
    check(_lib.lib().cc_embed_tokens_bwd(C.byref(gpt2.cfg), tokens.numel(), _p(dout), _p(tokens), _p(dw), _stream(gpt2.arena.device)), "cc_embed_tokens_bwd")
    return dw[:V]


def beam_buffers(device, samples: int, beam: int, V: int) -> tuple:
    """(next_tokens int32 (samples*beam,), src_rows int32 (samples*beam,), scratch) of cc_beam_step.  Owned by ONE decode (a
    generate_beam_tokens call allocates them once and reuses them from step to step: each step's outputs are consumed by
    cc_beam_advance before the next update) — never shared between decodes, streams or threads."""
    return (torch.empty(samples * beam, dtype=torch.int32, device=device), torch.empty(samples * beam, dtype=torch.int32, device=device),
            torch.empty(_lib.lib().cc_beam_ws_bytes(samples, beam, V), dtype=torch.uint8, device=device))


def beam_step(logits: torch.Tensor, samples: int, beam: int, temperature: float, first: bool, stop_token: int, scores: torch.Tensor,
              seq_lengths: torch.Tensor, has_stopped: torch.Tensor, bufs: Optional[tuple] = None, lpart: Optional[tuple] = None):
    """One device-side beam update for `samples` independent beam sets (reference inference/base.py:84-119).
    logits fp32 (samples*beam, V) (a view with row stride ldl is fine); state tensors are updated IN PLACE.
    ``bufs``: the caller's beam_buffers(...) (fresh ones are allocated when omitted).  ``lpart``: DecodeSession.lpart of the forward
    that produced these logits (softmax partials from the lm_head epilogue): the update then runs as one launch (cc_beam_step_p).
    Returns (next_tokens int32 (samples*beam,), src_rows int32 (samples*beam,) local row index inside each sample)."""
    dev = logits.device
    V = logits.shape[1]
    ldl = logits.stride(0)
    nt, sr, ws = bufs if bufs is not None else beam_buffers(dev, samples, beam, V)
    check(_lib.lib().cc_beam_step_p(samples, beam, V, _p(logits), ldl, _p(lpart[0]) if lpart is not None else None,
                                   lpart[1] if lpart is not None else 0, float(temperature), int(first), int(stop_token), _p(scores),
                                   _p(seq_lengths), _p(has_stopped), _p(nt), _p(sr), _p(ws), _stream(dev)), "cc_beam_step_p")
    return nt, sr


def sample_step(logits: torch.Tensor, u: torch.Tensor, temperature: float = 1.0, top_k: int = 0, top_p: float = 0.0, mode: int = 0,
                history: torch.Tensor = None, hist_len: int = 0, repetition_penalty: float = 1.0, return_probs: bool = False,
                length_penalty_stop: int = -1, length_penalty: float = 1.0):
    """One device-side sampling step for every row of fp32 ``logits`` (R, V) (reference inference/base.py:159-184 for mode 0 =
    generate_nucleus_sampling, :245-262 + utils.py:5-37 for mode 1 = top_k_top_p_filtering + softmax).  ``u`` (R,) uniforms in
    [0, 1) drive the inverse-CDF draw; ``history`` int64 (R, >= hist_len) feeds the repetition penalty and, with
    ``length_penalty_stop`` >= 0, the sentence-length penalty of no_beam.py:55-60 (history tokens whose filtered value equals
    float(stop id) are multiplied by ``length_penalty``, utils.py:40-51).
    Returns next_tokens int32 (R,) [, probs fp32 (R, V) — the pre-sampling distribution]."""
    dev = logits.device
    R, V = logits.shape
    nt = torch.empty(R, dtype=torch.int32, device=dev)
    probs = torch.empty(R, V, dtype=torch.float32, device=dev) if return_probs else None
    if history is not None:
        assert history.dtype == torch.int64 and history.shape[0] == R and history.stride(1) == 1
    check(_lib.lib().cc_sample_step_lp(_p(logits), R, V, logits.stride(0), float(temperature), int(top_k or 0), float(top_p or 0.0), int(mode),
                                      _p(history) if history is not None else None, int(hist_len),
                                      history.stride(0) if history is not None else 0, float(repetition_penalty), int(length_penalty_stop),
                                      float(length_penalty), _p(u), _p(nt), _p(probs) if probs is not None else None, _stream(dev)),
          "cc_sample_step_lp")
    return (nt, probs) if return_probs else nt
