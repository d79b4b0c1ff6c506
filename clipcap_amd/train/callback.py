"""Checkpoint / config I/O with the reference's file naming (clipcap/train/callback.py:5-28):
``<prefix>_config.yaml``, ``<prefix>_epoch_<n>.ckpt``, ``<prefix>_final.ckpt``; a ``.ckpt`` is a dict with ``"state_dict"``
(what ``load(..., from_checkpoint=True)`` reads, load.py:31-32) plus the optimizer state for true resume (which the reference lacks)."""
from __future__ import annotations

from pathlib import Path

import torch
import yaml


class CheckpointSaver:
    def __init__(self, output_path: str = "./checkpoints/", filename_prefix: str = "clipclap_demo", save_every_n_epochs: int = 1,
                 use_deepspeed: bool = False) -> None:
        self.output_path = Path(output_path)
        self.output_path.mkdir(parents=True, exist_ok=True)
        self.filename_prefix = filename_prefix
        self.save_every_n_epochs = max(1, int(save_every_n_epochs))
        self.use_deepspeed = use_deepspeed

    def save_config(self, config: dict) -> None:
        with open(self.output_path / f"{self.filename_prefix}_config.yaml", "w+") as f:
            yaml.dump(config, f, default_flow_style=False)

    @staticmethod
    def optimizer_state(model) -> dict:
        """AdamW moments per arena over the WHOLE arena.  With sharded optimizer state (ddp.ZeroShard) this is a collective: every rank
        calls it, rank 0 writes the result."""
        opt = {}
        for name, mod in (("mapper", model.transformer_mapper), ("lm", model.language_model)):
            a = mod.engine.arena
            if a.m is not None:
                m, v = a.full_moments()
                opt[name] = {"m": m.cpu(), "v": v.cpu()}
        return opt

    def _write(self, model, path: Path, extra: dict) -> None:
        state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        extra = dict(extra)
        opt = extra.pop("optimizer_state", None)
        if opt is None:
            opt = self.optimizer_state(model)
        sc = getattr(model.engine, "scaler", None)      # fp16 operands: [scale, good steps, applied steps] of the dynamic loss scale
        if sc is not None:
            extra = dict(extra, loss_scaler=sc.state.cpu())
        torch.save({"state_dict": state, "optimizer_state": opt, "optimizer_step": getattr(model, "_opt_step", 0), **extra}, path)

    def on_epoch_end(self, model, epoch: int, **extra) -> None:
        if epoch % self.save_every_n_epochs == 0:
            self._write(model, self.output_path / f"{self.filename_prefix}_epoch_{epoch}.ckpt", dict(epoch=epoch, **extra))

    def save_final_checkpoint(self, model, **extra) -> None:
        self._write(model, self.output_path / f"{self.filename_prefix}_final.ckpt", extra)


def resume(model, ckpt_path: str) -> dict:
    """Restores weights + AdamW moments + step counter from a checkpoint written above."""
    ck = torch.load(ckpt_path, map_location="cpu")
    model.load_state_dict(ck["state_dict"], strict=False)
    for name, mod in (("mapper", model.transformer_mapper), ("lm", model.language_model)):
        st = ck.get("optimizer_state", {}).get(name)
        if st is not None:
            a = mod.engine.arena
            a.m = st["m"].to(a.device)
            a.v = st["v"].to(a.device)
            if a.zero is not None:                      # sharding already configured: keep the own range only
                lo, hi = a.zero[1][a.zero[0]]
                a.m, a.v = a.m[lo:hi].clone(), a.v[lo:hi].clone()
    model._opt_step = int(ck.get("optimizer_step", 0))
    if "loss_scaler" in ck and model.language_model.engine.arena.device.type == "cuda":
        sc = model.engine._scaler(model.language_model.engine.arena.device)
        if sc is not None:      # resumed fp16 run: continue at the saved scale instead of restarting at 2^16 (and skipping steps again)
            st = ck["loss_scaler"].to(sc.state.device, torch.float32)
            sc.state[: st.numel()].copy_(st)
            if st.numel() < 3:      # checkpoint written before the applied-step counter existed: Adam's bias correction continues from the
                sc.state[2] = float(model._opt_step)     # saved optimizer step instead of restarting at t = 1
    return ck
