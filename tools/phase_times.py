"""[PPO_PRECISION=medium] Synchronised wall time of the two phases of a bench iteration (bench.py's own phase split is host time: the train
phase's kernels still run when its host code returns).  Usage: python tools/phase_times.py [n_steps] [iters]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_amd import envs, logger, models, rollout  # noqa: E402
from ppo_amd.config import args  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
args.setup(["--agents=256", f"--n_steps={N}", "--model_architecture=single", "--model_encoder=impala", "--env_type=synthetic",
            "--env_embed_time=False", "--seed=1", "--device=cuda", "--policy_opt_mini_batch_size=256", "--policy_opt_epochs=2",
            "--disable_logging=True", "--upload_batch=True", "--env_reward_normalization=off"])
torch.manual_seed(1)
np.random.seed(1)
shape, nA = envs.get_env_spec()
model = models.TVFModel("impala", input_dims=shape, actions=nA, device="cuda", architecture="single", hidden_units=256,
                        head_scale=0.1, head_bias=True, precision=os.environ.get("PPO_PRECISION", "high"))
r = rollout.Runner(model, logger.Logger(quiet=True))
r.vec_env = envs.create_envs_classic()
r.reset()
for _ in range(1):
    r.generate_rollout(); r.calculate_returns(); r.train()
torch.cuda.synchronize()
t = {"rollout": 0.0, "returns": 0.0, "train": 0.0}
for _ in range(iters):
    for name, fn in (("rollout", r.generate_rollout), ("returns", r.calculate_returns), ("train", r.train)):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        t[name] += time.perf_counter() - t0
n_mb = 2 * (N * 256 // 256)
print({k: round(v / iters * 1e3, 2) for k, v in t.items()}, "ms per iteration;",
      f"rollout {t['rollout'] / iters / (N + 1) * 1e3:.3f} ms per env step, train {t['train'] / iters / n_mb * 1e3:.3f} ms per minibatch")
