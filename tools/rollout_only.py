"""Only rollouts of the bench configuration (for a rocprofv3 --kernel-trace of the rollout's kernel timeline).
Usage: python tools/rollout_only.py [n_steps] [rollouts]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_amd import envs, logger, models, rollout  # noqa: E402
from ppo_amd.config import args  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
args.setup(["--agents=256", f"--n_steps={N}", "--model_architecture=single", "--model_encoder=impala", "--env_type=synthetic",
            "--env_embed_time=False", "--seed=1", "--device=cuda", "--policy_opt_mini_batch_size=256", "--policy_opt_epochs=2",
            "--disable_logging=True", "--upload_batch=True", "--env_reward_normalization=off",
            f"--env_synthetic_threads={os.environ.get('PPO_SYNTH_THREADS', '8')}"] + os.environ.get("PPO_EXTRA_ARGS", "").split())
torch.manual_seed(1)
np.random.seed(1)
shape, nA = envs.get_env_spec()
model = models.TVFModel("impala", input_dims=shape, actions=nA, device="cuda", architecture="single", hidden_units=256,
                        head_scale=0.1, head_bias=True)
r = rollout.Runner(model, logger.Logger(quiet=True))
r.vec_env = envs.create_envs_classic()
r.reset()
r.generate_rollout()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    r.generate_rollout()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"rollout {dt / reps / (N + 1) * 1e3:.3f} ms per env step ({reps} rollouts of {N} steps)")
