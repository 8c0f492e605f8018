from ppo_amd.wrappers import VecNormalizeRewardWrapper, VecRepeatedActionPenalty, VecWrapper, get_wrapper  # noqa: F401
from ppo_amd.env_wrappers import *  # noqa: F401,F403,E402  (the per-env gym wrappers of the make stacks)
