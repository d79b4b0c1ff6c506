"""``add_model_args`` — the reference's model flags with the same names, types and defaults (clipcap/model/args.py:3-47),
including its ``type=bool`` quirk (any non-empty string parses as True)."""
from argparse import ArgumentParser

_MODEL_FLAGS = [
    ("--language-model", str, "gpt2-xl", "Language model (local directory or hub name) the prefix is fed to."),
    ("--prefix-length", int, 10, "Number of learned prefix rows appended after the projected embedding rows."),
    ("--projection-length", int, 10, "Number of LM-width rows one encoder embedding is projected into."),
    ("--train-language-model", bool, False, "Finetune the language model together with the mapper."),
    ("--transformer-layers", int, 8, "Layers of the mapping transformer."),
    ("--transformer-attention-heads", int, 8, "Attention heads of the mapping transformer."),
    ("--use-positional-embeddings", bool, True, "Windowed embeddings: add learned positional embeddings in the mapper."),
]


def add_model_args(parser: ArgumentParser) -> ArgumentParser:
    group = parser.add_argument_group("model")
    for flag, typ, default, text in _MODEL_FLAGS:
        group.add_argument(flag, type=typ, default=default, help=text)
    return parser
