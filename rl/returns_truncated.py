from ppo_amd.returns_truncated import *  # noqa: F401,F403
from ppo_amd import returns_truncated as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
