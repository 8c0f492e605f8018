"""Checkpoint container — host mirror of the reference's checkpoint I/O (rl/rollout.py:394-517,
rl/utils.py:977-1038): one `torch.save`d dict, gzip-compressed when `--checkpoint_compression` (the
reference's default, file name + ".gz"), holding the model `state_dict` under the reference's key names, the
optimiser states, counters, and the env / wrapper state gathered by walking the wrapper chain.

Everything written is plain containers + tensors, so files load with `torch.load(..., weights_only=True)`:
nothing from a checkpoint is ever executed.  NumPy arrays, tuples and NumPy scalars (env state, RNG state,
running moments) are converted to tagged plain forms on the way out and back on the way in.
"""
import gzip
import os

import numpy as np
import torch


def to_plain(obj):
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu()
    if isinstance(obj, np.ndarray):
        if obj.dtype == object:
            raise TypeError("object arrays cannot be checkpointed")
        # torch has no uint16/32/64 tensors everywhere: carry the raw bytes
        return {"__nd__": str(obj.dtype), "shape": list(obj.shape),
                "data": torch.from_numpy(np.frombuffer(np.ascontiguousarray(obj).tobytes(), dtype=np.uint8).copy())}
    if isinstance(obj, np.generic):
        return {"__np__": str(obj.dtype), "v": obj.item()}
    if isinstance(obj, tuple):  # tuples survive a weights-only load as they are (Adam's `betas` keeps its type)
        return tuple(to_plain(x) for x in obj)
    if isinstance(obj, list):
        return [to_plain(x) for x in obj]
    if isinstance(obj, dict):  # keys: str, or int (torch.optim's per-parameter state is keyed by parameter index)
        return {k: to_plain(v) for k, v in obj.items()}
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    raise TypeError(f"cannot checkpoint a {type(obj).__name__}")


def from_plain(obj):
    if isinstance(obj, dict):
        if "__nd__" in obj:
            raw = obj["data"].numpy().tobytes()
            return np.frombuffer(raw, dtype=np.dtype(obj["__nd__"])).reshape(obj["shape"]).copy()
        if "__np__" in obj:
            return np.dtype(obj["__np__"]).type(obj["v"])
        if "__tuple__" in obj:
            return tuple(from_plain(x) for x in obj["__tuple__"])
        return {k: from_plain(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [from_plain(x) for x in obj]
    if isinstance(obj, tuple):
        return tuple(from_plain(x) for x in obj)
    return obj


def save_env_state(env):
    """{wrapper class name: its save_state dict} down the `.env` chain (rl/utils.py:1006-1036); the process
    pool contributes one `vec_XXX` entry per env."""
    out = {}
    while env is not None:
        if hasattr(type(env), "save_state"):
            buf = {}
            env.save_state(buf)
            if buf:
                out[type(env).__name__] = buf
        env = env.__dict__.get("env")
    return out


def restore_env_state(env, state):
    while env is not None:
        key = type(env).__name__
        if key in state and hasattr(type(env), "restore_state"):
            env.restore_state(state[key])
        env = env.__dict__.get("env")


def save(data: dict, filename: str, compress: bool):
    """Returns the path written (filename + '.gz' when compressed, as the reference)."""
    plain = to_plain(data)
    if compress:
        filename = filename + ".gz"
        with gzip.open(filename, "wb", compresslevel=5) as f:
            torch.save(plain, f)
    else:
        with open(filename, "wb") as f:
            torch.save(plain, f)
    return filename


def load(path: str, map_location="cpu"):
    """Accepts the name with or without '.gz' (rl/utils.py `open_checkpoint`)."""
    if not os.path.exists(path) and os.path.exists(path + ".gz"):
        path = path + ".gz"
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        return from_plain(torch.load(f, map_location=map_location, weights_only=True))
