"""``add_training_args`` — the reference's training / data / deepspeed / wandb flags with identical names, types and defaults
(clipcap/train/args.py:3-113).  Flags that configured Lightning/DeepSpeed are accepted for command-line compatibility;
``--deepspeed-strategy`` with a ZeRO stage >= 1 shards the AdamW moments over the ranks (train/ddp.py ZeroShard), stage >= 2 also reduces every
gradient slice onto its owner only (GradReducer.set_owners); multi-GPU is always one process per GPU + RCCL; ``--enable-deepspeed`` keeps its one numerical effect, the optimizer's weight decay (0.0 under DeepSpeed's FusedAdam, 0.01 under torch.optim.AdamW)."""
from argparse import ArgumentParser


def _precision(text: str):
    """the reference's integer values (clipcap/train/args.py:30-34) plus 'bf16', the throughput mode of this build"""
    return "bf16" if text.strip().lower() == "bf16" else int(text)

_GROUPS = {
    "training": [
        ("--batch-size", int, 64, "Samples per batch (per process)."),
        ("--epochs", int, 5, "Passes over the training data."),
        ("--optimizer-lr", float, 2e-5, "AdamW learning rate."),
        ("--scheduler-warmup-steps", int, 5000, "Linear warm-up length in optimizer steps."),
        ("--fp-precision", _precision, 32, "Floating point precision (16/32/64, as the reference; or bf16): 32 / 64 = split-bf16 MFMA operands (3 terms per "
         "product, fp32 activations: the reference's fp32 default to 1e-3 on the logits, ~1/3 of the GEMM rate), 16 = fp16 operands + dynamic "
         "loss scaling, bf16 = bf16 operands (fastest).  fp32 accumulation and master weights in all of them."),
        ("--checkpoint-save-frequency", int, 1, "Write a checkpoint every n epochs."),
        ("--checkpoint-filename-prefix", str, 1, "Checkpoint file name prefix."),
        ("--device", str, "0", "GPU index, comma list, or -1 for all (one process per GPU via torchrun)."),
        ("--resume-from", str, None, "Checkpoint (.ckpt written by this trainer) to resume from: weights, AdamW state, step, schedule, epoch."),
    ],
    "data": [
        ("--input-dataset", str, "./dataset/", "Preprocessed dataset folder (embeddings/*.npy, captions/*.parquet, encoder_config.yaml)."),
        ("--output-folder", str, "./models/", "Where checkpoints and the model config are written."),
        ("--reader-max-piece-size", int, 50, "MB of embeddings one reader worker loads (and tokenises) at a time (embedding-reader's max_piece_size)."),
        ("--reader-parallel-pieces", int, 10, "Pieces read / tokenised concurrently by background workers, delivered in order (embedding-reader's parallel_pieces; 0 = on the training thread)."),
    ],
    "deepspeed": [
        ("--enable-deepspeed", bool, False, "No DeepSpeed here (one process per GPU + RCCL always); kept for its one numerical effect in the reference: "
         "FusedAdam's default weight decay 0.0 instead of torch AdamW's 0.01 (clipcap/model/model.py:72-77)."),
        ("--deepspeed-strategy", str, None, "ZeRO stage by Lightning's names (deepspeed_stage_1 / _2 / _3 ...): any stage >= 1 shards the AdamW "
         "moments over the ranks (each rank steps its own slice of the flat parameter arena, the slices are broadcast back); stage 2 / 3 (and plain "
         "'deepspeed') also partition the gradients on the wire: a slice is summed only onto the rank that owns it (reduce instead of all-reduce). "
         "The flat parameter / gradient arenas stay resident on every rank (the kernels read and write them)."),
    ],
    "wandb": [
        ("--enable-wandb", bool, False, "Log the loss to Weights & Biases if the package is installed."),
        ("--wandb-project", str, "clipcap", "W&B project name."),
        ("--logging-frequency", int, 50, "Log every n steps."),
    ],
}


def add_training_args(parser: ArgumentParser) -> ArgumentParser:
    for title, flags in _GROUPS.items():
        group = parser.add_argument_group(title)
        for flag, typ, default, text in flags:
            group.add_argument(flag, type=typ, default=default, help=text)
    return parser
