"""Drop-in module name of the reference (rl/procgen.py): the implementation lives in ppo_amd.procgen."""
from ppo_amd.procgen import *  # noqa: F401,F403
from ppo_amd.procgen import make  # noqa: F401
