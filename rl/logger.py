from ppo_amd.logger import *  # noqa: F401,F403
from ppo_amd import logger as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
