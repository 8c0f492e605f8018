from ppo_amd.vec_env import *  # noqa: F401,F403
from ppo_amd.vec_env import SyntheticVecEnv  # noqa: F401
