"""Drop-in module name of the reference (rl/atari.py): the implementation lives in ppo_amd.atari."""
from ppo_amd.atari import *  # noqa: F401,F403
from ppo_amd.atari import make  # noqa: F401
