"""Environment factory — host mirror of the reference's rl/envs.py (`create_envs_classic` :135-172,
`make_env` :16-28).  gym / ALE / procgen / mujoco are not installed in this build's image, so the env
families constructed here are `synthetic` (the benchmark workload) and `classic`
(built-in CartPole, BASELINE.json configs[0]); for the real families the factory takes a list of
user-supplied `env_fns` (gym API).  gym-API envs run in the process pool and get the reference's
vector wrappers (reward normalisation, repeated-action penalty) in the reference's order.  Seeds
follow the reference: env i gets base_seed + i*997 with i the GLOBAL env index (rl/envs.py:146)."""
import functools

import numpy as np

from . import parallel
from .config import args
from .vec_env import SplitVecEnv, SyntheticVecEnv

OBS_SHAPES = {"atari": (4, 84, 84), "procgen": (3, 64, 64), "synthetic": (4, 84, 84), "classic": (4,)}


def get_env_spec():
    """(obs_shape, n_actions) for the configured env family."""
    t = args.env.type
    shape = OBS_SHAPES.get(t, (4, 84, 84))
    if t == "atari" and args.env.embed_time:
        shape = (5, 84, 84)  # FrameStack then TimeChannelWrapper (rl/atari.py:217-220)
    if t == "procgen" and args.env.embed_time:
        shape = (4, 64, 64)
    n_actions = {"atari": 6, "procgen": 15, "synthetic": 6, "classic": 2}.get(t, 6)
    if t == "synthetic":  # the benchmark workload in the shape of any config (BASELINE.json configs[1..3])
        if args.env.synthetic_shape:
            shape = tuple(int(v) for v in str(args.env.synthetic_shape).split(","))
        if args.env.synthetic_actions > 0:
            n_actions = int(args.env.synthetic_actions)
    return shape, n_actions


def _classic_env_fns(N, base_seed, first):
    from . import classic_envs
    if args.env.name.lower().startswith("cartpole"):
        return [functools.partial(classic_envs.make_cartpole, base_seed + (first + i) * 997) for i in range(N)]
    raise ValueError(f"no built-in classic env named '{args.env.name}' (CartPole is built in; pass env_fns for others)")


def _moments_sync(moments):
    """Sum the reward normaliser's per-step moments over data-parallel ranks."""
    import torch
    t = torch.from_numpy(np.asarray(moments, np.float64).copy())
    if parallel.world_size() > 1:
        on_gpu = parallel.backend_name() == "nccl"  # RCCL reduces device buffers; gloo (CPU tests) host ones
        t = t.cuda() if on_gpu else t
        parallel.allreduce_sum_(t)
        t = t.cpu()
    return t.numpy()


def create_envs_classic(N=None, rank=0, world=1, env_fns=None, monitor_video=False):
    N = N or args.agents
    base_seed = args.seed if args.seed is not None and args.seed >= 0 else 0
    if env_fns is None and args.env.type == "classic":
        env_fns = _classic_env_fns(N, base_seed, rank * N)
    if env_fns is not None:
        from . import wrappers
        from .hybrid_vec_env import HybridAsyncVectorEnv
        workers = args.workers if args.workers > 0 else min(8, len(env_fns))
        while len(env_fns) % workers:
            workers -= 1
        vec_env = HybridAsyncVectorEnv(env_fns, copy=False, max_cpus=workers)
        vec_env.pin()  # async H2D straight out of the shared observation block
        if args.env.reward_normalization == "rms":
            vec_env = wrappers.VecNormalizeRewardWrapper(
                vec_env, gamma=args.reward_normalization_gamma, mode="rms", clip=args.env.reward_normalization_clipping,
                moments_sync=_moments_sync if world > 1 else None)
        if args.env.max_repeated_actions > 0 and args.env.type != "mujoco":
            vec_env = wrappers.VecRepeatedActionPenalty(vec_env, args.env.max_repeated_actions,
                                                        args.env.repeated_action_penalty)
        return vec_env

    shape, n_actions = get_env_spec()

    def make(n, offset):
        return SyntheticVecEnv(n, obs_shape=shape, n_actions=n_actions, seed=base_seed,
                               p_done=args.env.synthetic_done_prob, env_offset=offset,
                               threads=args.env.synthetic_threads)

    parts = int(args.env.pipeline_parts)
    if parts > 1 and N % parts == 0 and N // parts >= 16:
        # env streams are keyed by the GLOBAL env index, so the split changes nothing but the overlap
        per = N // parts
        return SplitVecEnv([make(per, rank * N + i * per) for i in range(parts)])
    return make(N, rank * N)
