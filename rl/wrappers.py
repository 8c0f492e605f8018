from ppo_amd.wrappers import VecNormalizeRewardWrapper, VecRepeatedActionPenalty, VecWrapper, get_wrapper  # noqa: F401
