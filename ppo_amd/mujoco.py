"""MuJoCo env construction — host mirror of the reference's rl/mujoco.py `make` (:29-67).  The simulator comes from
`gym.make("<id>-v2")` when gym + mujoco_py are installed (not part of this build's image), or from `base_env`
(classic gym API, flat float observations, continuous actions)."""
import numpy as np

from . import env_wrappers as W
from .config import args as global_args


def make(env_id: str, monitor_video=False, seed=None, args=None, determanistic_saving=True, base_env=None):
    args = args or global_args
    e = args.env
    assert e.frame_skip == 1, "Frame skip should be 1 for mujoco"
    if base_env is None:
        try:
            import gym
        except ImportError as err:
            raise ImportError("gym + mujoco_py are needed to create MuJoCo envs (or pass base_env=...)") from err
        base_env = gym.make(f"{env_id}-v2")
    env = W.LabelEnvWrapper(base_env, "env_id", env_id)
    if seed is not None:
        np.random.seed(seed)
        env.seed(seed)
    if e.timeout > 0:
        env = W.TimeLimitWrapper(env, e.timeout)
    if e.embed_time:
        env = W.TimeFeatureWrapper(env)
    env = W.F32Wrapper(env)
    env = W.EpisodeScoreWrapper(env)
    return W.MonitorWrapper(env, monitor_video=False)
