"""Environment factory — host mirror of the reference's rl/envs.py (`create_envs_classic` :135-172,
`make_env` :16-28).  gym / ALE / procgen / mujoco are not installed in this build's image, so the only
env family constructed here is `synthetic`; for the real families the factory takes a
list of user-supplied `env_fns` (gym API) and returns the process-pool vector env.  Seeds follow the
reference: env i gets base_seed + i*997 with i the GLOBAL env index (rl/envs.py:146)."""
from .config import args
from .vec_env import SplitVecEnv, SyntheticVecEnv

OBS_SHAPES = {"atari": (4, 84, 84), "procgen": (3, 64, 64), "synthetic": (4, 84, 84)}


def get_env_spec():
    """(obs_shape, n_actions) for the configured env family on synthetic data."""
    t = args.env.type
    shape = OBS_SHAPES.get(t, (4, 84, 84))
    if t == "atari" and args.env.embed_time:
        shape = (5, 84, 84)  # FrameStack then TimeChannelWrapper (rl/atari.py:217-220)
    if t == "procgen" and args.env.embed_time:
        shape = (4, 64, 64)
    n_actions = {"atari": 6, "procgen": 15, "synthetic": 6}.get(t, 6)
    return shape, n_actions


def create_envs_classic(N=None, rank=0, world=1, env_fns=None):
    N = N or args.agents
    if env_fns is not None:
        from .hybrid_vec_env import HybridAsyncVectorEnv
        return HybridAsyncVectorEnv(env_fns, max_cpus=args.workers if args.workers > 0 else 8)
    shape, n_actions = get_env_spec()
    base_seed = args.seed if args.seed >= 0 else 0


    def make(n, offset):
        return SyntheticVecEnv(n, obs_shape=shape, n_actions=n_actions, seed=base_seed,
                               p_done=args.env.synthetic_done_prob, env_offset=offset,
                               threads=args.env.synthetic_threads)

    parts = int(getattr(args.env, "pipeline_parts", 2))
    if parts > 1 and N % parts == 0 and N // parts >= 16:
        # env streams are keyed by the GLOBAL env index, so the split changes nothing but the overlap
        per = N // parts
        return SplitVecEnv([make(per, rank * N + i * per) for i in range(parts)])
    return make(N, rank * N)
