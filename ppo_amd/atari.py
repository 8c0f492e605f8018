"""Atari env construction — host mirror of the reference's rl/atari.py `make` (:119-230): the wrapper stack between
ALE and the vector env.  The simulator comes from `gym.make("ALE/<id>-v5", obs_type="rgb", frameskip=1, ...)` when
gym + ale-py are installed (they are not part of this build's image), or from `base_env` — anything with the classic
gym API that yields 210x160x3 uint8 frames — which is how the stack is tested.  The ROM MD5 table (rl/atari.py:12-117)
is not carried: `--env_atari_rom_check` is accepted and ignored."""
import numpy as np

from . import env_wrappers as W
from .config import Config, args as global_args


def make(env_id: str, monitor_video=False, seed=None, args=None, determanistic_saving=True, base_env=None):
    args = args or global_args
    e = args.env
    if base_env is None:
        try:
            import gym
        except ImportError as err:
            raise ImportError("gym + ale-py are needed to create Atari envs (or pass base_env=...)") from err
        base_env = gym.make(f"ALE/{env_id}-v5", obs_type="rgb", frameskip=1,  # our FrameSkipWrapper applies the max
                            repeat_action_probability=e.repeat_action_probability,
                            full_action_space=e.full_action_space).unwrapped
    env = W.LabelEnvWrapper(base_env, "env_id", env_id)
    if seed is not None:
        env = W.LabelEnvWrapper(env, "seed", seed)
        np.random.seed(seed)
        env.seed(seed)
    if e.per_step_termination_probability > 0:
        env = W.RandomTerminationWrapper(env, e.per_step_termination_probability)
    env = W.SaveEnvStateWrapper(env, determanistic=determanistic_saving)
    if e.noop_duration > 0:
        env = W.NoopResetWrapper(env, noop_max=e.noop_duration)
    env = W.FrameSkipWrapper(env, min_skip=e.frame_skip, max_skip=e.frame_skip, reduce_op=np.max)
    if e.timeout > 0:
        env = W.TimeLimitWrapper(env, e.timeout)
    if env_id == "MontezumaRevenge":
        env = W.MontezumaInfoWrapper(env)  # after the frame skip: rooms are those of the most recent frame
    env = W.MonitorWrapper(env, monitor_video=monitor_video)
    env = W.EpisodeScoreWrapper(env)
    if e.reward_clipping == "sqrt":
        env = W.SqrtRewardWrapper(env)
    elif e.reward_clipping != "off":
        try:
            clip = float(e.reward_clipping)
        except (TypeError, ValueError):
            raise ValueError("reward_clipping should be off, sqrt, or a float")
        env = W.ClipRewardWrapper(env, clip)
    res = Config.RESOLUTIONS[e.resolution][0]  # the reference's res_x and res_y both read element 0 (rl/config.py:543-548)
    env = W.AtariWrapper(env, width=res, height=res)
    if e.zero_obs:
        env = W.ZeroObsWrapper(env)
    env = W.ColorTransformWrapper(env, e.color_mode)
    if e.atari_terminal_on_loss_of_life:
        env = W.EpisodicLifeEnv(env)
    if e.deferred_rewards != 0:
        env = W.DeferredRewardWrapper(env, e.deferred_rewards)
    if e.embed_action:
        env = W.ActionAwareWrapper(env)  # before the frame stack: every stacked frame carries its own action
    env = W.FrameStack(env, n_stacks=e.frame_stack)
    if e.embed_time:
        env = W.TimeChannelWrapper(env)
    if e.embed_state:
        env = W.StateHistoryWrapper(env)
    env = W.ChannelsFirstWrapper(env)
    return W.NullActionWrapper(env)
