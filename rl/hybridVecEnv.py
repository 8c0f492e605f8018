from ppo_amd.hybrid_vec_env import HybridAsyncVectorEnv  # noqa: F401
from ppo_amd.vec_env import SplitVecEnv, SyntheticVecEnv  # noqa: F401
