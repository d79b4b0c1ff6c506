"""``train(args)`` / ``start_training()`` — the driver of ``python -m clipcap_amd.train`` (reference clipcap/train/train.py:17-104),
without Lightning/DeepSpeed: one process per GPU (launch N>1 with ``python -m torch.distributed.run --nproc-per-node N -m
clipcap_amd.train ...``), rank-sharded batches, fused forward+backward+AdamW steps, RCCL all-reduce of the flat gradient
arenas with a global kept-token divisor (clipcap_amd/train/ddp.py)."""
from __future__ import annotations

import os
from argparse import ArgumentDefaultsHelpFormatter, ArgumentParser, Namespace
from pathlib import Path

import torch
import yaml

from clipcap_amd.encoders.config import EncoderConfig
from clipcap_amd.model import ClipCapModel, ClipCapModelPrefixOnly, Config, TrainingConfig, add_model_args
from clipcap_amd.model.optim import linear_warmup_decay
from clipcap_amd.train.args import add_training_args
from clipcap_amd.train.callback import CheckpointSaver, resume
from clipcap_amd.train.dataloader import DevicePrefetcher, get_dataloader
from clipcap_amd._lib import OP_BF16, OP_FP16, OP_X3
from clipcap_amd.train.ddp import GradReducer, ZeroShard, zero_stage


def grad_wire_dtype(op_dtype: int, train_lm: bool) -> torch.dtype:
    """Element type of the gradient all-reduce.  The split-bf16 mode exists to reproduce the reference's fp32 arithmetic, so its gradients
    travel in fp32 — N-rank gradients then equal 1-rank gradients to fp32 rounding, as in the reference's DDP (train.py:77-85).  In the
    bf16 / fp16 throughput modes a frozen-LM run sends bf16 (only the mapper's 41.7 M gradients travel and only the short mapper backward
    can hide them; the fp32 arena stays the accumulator; tests/test_ddp_gloo.py: gradients to 8e-3, 20-step loss trajectory to 2e-3); a
    full finetune sends fp32."""
    if op_dtype == OP_X3 or train_lm:
        return torch.float32
    return torch.bfloat16


def train(args: Namespace, tokenizer=None, language_model=None, step_hook=None) -> int:
    """``step_hook(step, model)`` (optional, not in the reference): called after every optimizer step has been enqueued — a measurement hook
    (bench.py --mode e2e times the loop with it), never used by ``python -m clipcap_amd.train``."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    first = str(args.device).split(",")[0]
    dev_index = local if world > 1 or first == "-1" else int(first)
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)
    if world > 1 and not torch.distributed.is_initialized():
        torch.distributed.init_process_group("nccl", device_id=device)          # RCCL over xGMI

    with open(Path(args.input_dataset) / "encoder_config.yaml", "r") as f:      # train.py:26-29
        encoder_config = EncoderConfig(**yaml.safe_load(f))
    dataset, encoder_embedding_size = get_dataloader(args.input_dataset, args.language_model, args.batch_size, tokenizer=tokenizer,
                                                     rank=rank, world_size=world,
                                                     reader_max_piece_size=getattr(args, "reader_max_piece_size", 50),
                                                     reader_parallel_pieces=getattr(args, "reader_parallel_pieces", 10),
                                                     max_token_length=getattr(args, "max_token_length", 64))
    encoder_config.encoder_embedding_size = encoder_embedding_size
    args.total_steps = len(dataset) * args.epochs                                # train.py:40
    config = Config.from_args(args)
    config.training_config = TrainingConfig.from_args(args)
    config.encoder_config = encoder_config
    cls = ClipCapModel if args.train_language_model else ClipCapModelPrefixOnly  # train.py:46-50
    model = cls(config, language_model=language_model).set_precision(args.fp_precision).to(device)   # train.py:82 precision=
    model.train()
    step, first_epoch = 0, 0
    if getattr(args, "resume_from", None):
        # true resume (the reference has none, train.py:17-93): weights, AdamW moments, optimizer step, schedule position, epoch
        ck = resume(model, args.resume_from)
        step = int(ck.get("step", ck.get("optimizer_step", 0)))
        first_epoch = int(ck["epoch"]) + 1 if "epoch" in ck else step // max(1, len(dataset))
    if world > 1:
        # every rank must start from the same parameters (and, on resume, the same moments): rank 0's are authoritative
        for mod in (model.transformer_mapper, model.language_model):
            a = mod.engine.arena
            torch.distributed.broadcast(a.w32, src=0)
            a.mark_dirty()                # the collective wrote through a raw pointer: the operand copy is rebuilt before its next use
            if a.m is not None:
                torch.distributed.broadcast(a.m, src=0)
                torch.distributed.broadcast(a.v, src=0)

    saver = CheckpointSaver(args.output_folder, args.checkpoint_filename_prefix, save_every_n_epochs=args.checkpoint_save_frequency)
    if rank == 0:
        saver.save_config(config.to_dict())                                      # train.py:60-68
    arenas = [model.transformer_mapper.engine.arena] + ([model.language_model.engine.arena] if model._train_lm else [])
    wire = grad_wire_dtype(model.transformer_mapper.engine.op_dtype, model._train_lm)
    if rank == 0:
        mode = {OP_BF16: "bf16 operands (throughput mode; --fp-precision bf16)", OP_FP16: "fp16 operands with dynamic loss-scaling (--fp-precision 16)",
                OP_X3: "split-bf16 operands = the reference's fp32 default (--fp-precision 32/64; ~2.3x the bf16 step time, logits within 1e-3)"}
        print(f"clipcap_amd: operand mode: {mode[model.transformer_mapper.engine.op_dtype]}; gradient wire: {str(wire).replace('torch.', '')}"
              + (f" over {world} ranks" if world > 1 else ""), flush=True)
    reducer = GradReducer([a.grads() for a in arenas], wire_dtype=wire) if world > 1 else None
    stage = zero_stage(getattr(args, "deepspeed_strategy", None)) if world > 1 else 0
    sharded = stage > 0
    if sharded:
        # --deepspeed-strategy (args.py:87-92): AdamW moments sharded over the ranks (ZeRO stage 1); after the resume broadcast above, so
        # that full moments of a resumed run are cut down to each rank's own range.  Stage 2 / 3: the gradient slices are then reduced
        # onto their owners only (half an all-reduce's bytes); the arenas themselves stay whole — the kernels read and write them
        owners = ZeroShard(rank, world).apply(arenas)
        if stage >= 2:
            reducer.set_owners(owners, rank)
        if rank == 0:
            print(f"clipcap_amd: optimizer state sharded over {world} ranks" + (", gradients reduced onto their owners" if stage >= 2 else "")
                  + f" (--deepspeed-strategy {args.deepspeed_strategy})", flush=True)
    sched = linear_warmup_decay(args.scheduler_warmup_steps, args.total_steps)
    logger = None
    if args.enable_wandb and rank == 0:
        try:
            import wandb
            logger = wandb.init(project=args.wandb_project)
        except ImportError:
            logger = None
    try:
        step = _run_epochs(args, model, dataset, device, sched, reducer, saver, sharded, rank, logger, first_epoch, step, step_hook)
    finally:
        if hasattr(dataset, "close"):
            dataset.close()          # the reader's worker processes (do not leave them to interpreter exit)
    opt = CheckpointSaver.optimizer_state(model) if sharded else None
    if rank == 0:
        saver.save_final_checkpoint(model, step=step, **({"optimizer_state": opt} if opt is not None else {}))
    return 0


def _run_epochs(args, model, dataset, device, sched, reducer, saver, sharded, rank, logger, first_epoch, step, step_hook):
    for epoch in range(first_epoch, args.epochs):
        for batch in DevicePrefetcher(dataset, device):
            loss = model.fused_step(batch, lr=args.optimizer_lr * sched(step), reducer=reducer)
            step += 1
            if step_hook is not None:
                step_hook(step, model)
            if rank == 0 and step % max(1, args.logging_frequency) == 0:
                val = float(loss)
                print(f"epoch {epoch} step {step}/{args.total_steps} loss {val:.4f}", flush=True)
                if logger is not None:
                    logger.log({"loss": val}, step=step)
        # with sharded optimizer state, collecting the moments is a collective: every rank takes part, rank 0 writes
        opt = CheckpointSaver.optimizer_state(model) if sharded and epoch % saver.save_every_n_epochs == 0 else None
        if rank == 0:
            saver.on_epoch_end(model, epoch, step=step, **({"optimizer_state": opt} if opt is not None else {}))
    return step


def start_training() -> int:
    parser = ArgumentParser(description=__doc__, formatter_class=ArgumentDefaultsHelpFormatter)
    parser = add_training_args(parser)
    parser = add_model_args(parser)
    return train(parser.parse_args())


if __name__ == "__main__":
    raise SystemExit(start_training())
