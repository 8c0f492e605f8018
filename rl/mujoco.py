"""Drop-in module name of the reference (rl/mujoco.py): the implementation lives in ppo_amd.mujoco."""
from ppo_amd.mujoco import *  # noqa: F401,F403
from ppo_amd.mujoco import make  # noqa: F401
