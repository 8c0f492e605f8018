"""Environment factory — host mirror of the reference's rl/envs.py (`create_envs_classic` :135-172,
`make_env` :16-28).  The env families are `atari`, `procgen`, `mujoco` (the reference's wrapper stacks,
ppo_amd/atari.py / procgen.py / mujoco.py, over simulators from gym — which this build's image does not
have: without it those three raise at construction and say so), `synthetic` (the benchmark workload,
SURVEY.md §8d) and `classic` (built-in CartPole, BASELINE.json configs[0]); a list of user-supplied
`env_fns` (gym API) replaces the family's own constructors.  gym-API envs run in the process pool and get the reference's
vector wrappers (reward normalisation, repeated-action penalty) in the reference's order.  Seeds
follow the reference: env i gets base_seed + i*997 with i the GLOBAL env index (rl/envs.py:146)."""
import functools

import numpy as np

from . import parallel
from .config import args
from .vec_env import EnvGroup, SplitVecEnv, SyntheticVecEnv

OBS_SHAPES = {"atari": (4, 84, 84), "procgen": (3, 64, 64), "synthetic": (4, 84, 84), "classic": (4,)}


def make_env(env_type, env_id, **kwargs):
    """One wrapped env of the given family (rl/envs.py:16-28)."""
    if env_type == "atari":
        from . import atari as family
    elif env_type == "mujoco":
        from . import mujoco as family
    elif env_type == "procgen":
        from . import procgen as family
    else:
        raise ValueError(f"Invalid environment type {env_type}")
    return family.make(env_id, **kwargs)


def _gym_available():
    import importlib.util
    return importlib.util.find_spec("gym") is not None


def get_env_spec():
    """(obs_shape, n_actions) for the configured env family: what the wrapper stack of that family produces
    (the reference builds one throw-away env to read its spaces, train.py:41; with gym installed so does this)."""
    t = args.env.type
    if t in ("atari", "procgen", "mujoco") and _gym_available():
        env = make_env(t, args.env.name, args=args)
        space = env.action_space
        n = space.n if hasattr(space, "n") and space.n is not None else int(space.shape[0])
        return tuple(env.observation_space.shape), int(n)
    shape = OBS_SHAPES.get(t, (4, 84, 84))
    if t == "atari":
        from .config import Config
        res = Config.RESOLUTIONS[args.env.resolution][0]
        per_frame = 1 if args.env.color_mode == "bw" else 3
        # FrameStack, then TimeChannelWrapper (rl/atari.py:215-220)
        shape = (args.env.frame_stack * per_frame + (1 if args.env.embed_time else 0), res, res)
    if t == "procgen":
        shape = (3 + (1 if args.env.embed_time else 0), 64, 64)
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
    if env_fns is None and args.env.type in ("atari", "procgen", "mujoco"):
        if not _gym_available():
            raise RuntimeError(f"env family '{args.env.type}' needs gym and its simulator package, which are not installed; "
                               "available without them: --env_type=synthetic | classic, or pass env_fns")
        # seeds by GLOBAL env index (rl/envs.py:146); args travels with the constructor: workers are spawned
        env_fns = [functools.partial(make_env, args.env.type, env_id=args.env.name, args=args,
                                     seed=base_seed + (rank * N + i) * 997, monitor_video=monitor_video) for i in range(N)]
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
        leaves = int(args.env.pipeline_leaves)
        if leaves > 1 and per % leaves == 0 and per // leaves >= 16:
            sub = per // leaves
            return SplitVecEnv([EnvGroup([make(sub, rank * N + i * per + k * sub) for k in range(leaves)])
                                for i in range(parts)])
        return SplitVecEnv([make(per, rank * N + i * per) for i in range(parts)])
    return make(N, rank * N)
