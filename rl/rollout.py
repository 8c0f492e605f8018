from ppo_amd.rollout import *  # noqa: F401,F403
from ppo_amd import rollout as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
