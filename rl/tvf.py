from ppo_amd.tvf import *  # noqa: F401,F403
from ppo_amd import tvf as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
