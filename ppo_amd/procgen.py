"""Procgen env construction — host mirror of the reference's rl/procgen.py `make` (:33-84).  The simulator comes from
`gym.make("procgen:procgen-<id>-v0", distribution_mode=..., rand_seed=...)` when gym + procgen are installed (not part
of this build's image), or from `base_env` (classic gym API, 64x64x3 uint8 frames)."""
import numpy as np

from . import env_wrappers as W
from .config import args as global_args


def make(env_id: str, monitor_video=False, seed=None, args=None, base_env=None):
    args = args or global_args
    e = args.env
    assert e.frame_skip == 1, "Frame skip should be 1 for procgen"
    if seed is not None:
        np.random.seed(seed)
    if base_env is None:
        try:
            import gym
        except ImportError as err:
            raise ImportError("gym + procgen are needed to create Procgen envs (or pass base_env=...)") from err
        kwargs = {"distribution_mode": e.procgen_difficulty}
        if seed is not None:
            kwargs["rand_seed"] = seed
        base_env = gym.make(f"procgen:procgen-{env_id}-v0", **kwargs)
    env = W.LabelEnvWrapper(base_env, "env_id", env_id)
    if e.timeout > 0:
        env = W.TimeLimitWrapper(env, e.timeout)
    env = W.ProcGenWrapper(env)
    env = W.MonitorWrapper(env, monitor_video=monitor_video)
    env = W.ColorTransformWrapper(env, e.color_mode)
    if e.embed_time:
        env = W.TimeChannelWrapper(env)
    env = W.ChannelsFirstWrapper(env)
    env = W.EpisodeScoreWrapper(env)
    if e.embed_action:
        env = W.ActionAwareWrapper(env)  # on the CHW frame: the block goes onto every channel
    return W.NullActionWrapper(env)
