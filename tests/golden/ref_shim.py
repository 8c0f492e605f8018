"""Import shim for running the REFERENCE (/root/reference, Python) in the build container.

Build-side tooling for generating golden vectors only — nothing here is reference code and
nothing here runs on the GPU box (the reference never travels).  The reference targets
numpy 1.22 / gym 0.21 / torch 1.12; this container has numpy 2.2 / torch 2.10 and no gym, cv2,
lz4, blosc or torchvision, so:
  * numpy aliases removed in numpy >= 1.24 are restored (np.bool, np.object, np.float, np.int, np.long);
  * the absent third-party modules are replaced by MagicMock modules (with real placeholder
    classes where the reference subclasses them);
  * rl.config.args is set up from an explicit argv.
Usage:  from ref_shim import load_reference; rl = load_reference(["--flag=value", ...])
"""
import sys
import types
from unittest import mock

import numpy as np

REF_ROOT = "/root/reference"


def _install_mocks():
    for name, val in (("bool", bool), ("object", object), ("float", float), ("int", int), ("long", np.int64)):
        if not hasattr(np, name):
            setattr(np, name, val)
    names = ["gym", "gym.version", "gym.wrappers", "gym.vector", "gym.vector.utils", "gym.vector.async_vector_env",
             "gym.envs", "gym.envs.atari", "gym.utils", "gym.envs.registration", "gym.spaces", "cv2", "torchvision",
             "lz4", "lz4.frame", "blosc"]
    for n in names:
        if n not in sys.modules:
            m = mock.MagicMock(name=n)
            m.__name__ = n
            m.__path__ = []
            sys.modules[n] = m
    gym = sys.modules["gym"]

    class _Wrapper:  # placeholder base class so `class X(gym.Wrapper)` resolves; forwards like gym.Wrapper does
        def __init__(self, env=None, *a, **k):
            self.env = env

        def __getattr__(self, name):  # only reached for attributes the wrapper itself lacks
            if name.startswith("_") or name == "env":
                raise AttributeError(name)
            return getattr(self.__dict__["env"], name)

        @property
        def unwrapped(self):
            env = self.__dict__.get("env")
            return env.unwrapped if hasattr(env, "unwrapped") else env

        def reset(self, **kwargs):
            return self.env.reset(**kwargs)

        def step(self, action):
            return self.env.step(action)

    class _Box:  # gym.spaces.Box / Discrete stand-ins: the reference reads .shape / .dtype / .n only
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape) if shape is not None else None, np.dtype(dtype)

    class _Discrete:
        def __init__(self, n):
            self.n = int(n)

    class _VecEnv:
        def __init__(self, *a, **k):
            pass

    gym.Wrapper = _Wrapper
    gym.ObservationWrapper = _Wrapper
    gym.RewardWrapper = _Wrapper
    gym.ActionWrapper = _Wrapper
    gym.Env = _Wrapper
    gym.vector = sys.modules["gym.vector"]
    gym.vector.SyncVectorEnv = _VecEnv
    gym.vector.AsyncVectorEnv = _VecEnv
    gym.vector.VectorEnv = _VecEnv
    sys.modules["gym.vector.async_vector_env"].AsyncVectorEnv = _VecEnv
    gym.version = sys.modules["gym.version"]
    gym.version.VERSION = "0.21.0"
    gym.spaces = sys.modules["gym.spaces"]
    gym.spaces.Box = _Box
    gym.spaces.Discrete = _Discrete


def load_reference(argv):
    """Returns the imported reference package `rl` with rl.config.args set up from argv."""
    _install_mocks()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    old = sys.argv
    sys.argv = ["train.py"] + list(argv)
    try:
        import rl  # noqa
        import rl.config
        rl.config.args.setup()
    finally:
        sys.argv = old
    return rl
