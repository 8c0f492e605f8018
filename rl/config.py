from ppo_amd.config import *  # noqa: F401,F403
from ppo_amd import config as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
