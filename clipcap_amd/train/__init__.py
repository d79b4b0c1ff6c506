"""clipcap_amd.train — mirrors clipcap/train/__init__.py:1-2 (train, start_training, add_training_args)."""


def __getattr__(name):
    if name in ("train", "start_training"):
        from clipcap_amd.train import train as _t
        return getattr(_t, name)
    if name == "add_training_args":
        from clipcap_amd.train.args import add_training_args
        return add_training_args
    raise AttributeError(name)
