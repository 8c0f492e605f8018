"""Interleaved A/B of training-phase switches inside ONE process: one rollout, then Runner.train() (2 epochs x 256
minibatches of the bench configuration) alternating the variants (separate processes differ by more than most switches
move; here the spread is ~0.1 %).
Usage: PPO_AB="models.ADAM_SCATTER=0,1" python tools/train_ab.py [rounds]      (module.ATTRIBUTE=value,value,...)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_amd import envs, logger, models, rollout  # noqa: E402
from ppo_amd.config import args  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
args.setup(["--agents=256", "--n_steps=256", "--model_architecture=single", "--model_encoder=impala", "--env_type=synthetic",
            "--env_embed_time=False", "--seed=1", "--device=cuda", "--policy_opt_mini_batch_size=256", "--policy_opt_epochs=2",
            "--disable_logging=True", "--upload_batch=True", "--env_reward_normalization=off"]
           + os.environ.get("PPO_EXTRA_ARGS", "").split())
torch.manual_seed(1)
np.random.seed(1)
shape, nA = envs.get_env_spec()
model = models.TVFModel("impala", input_dims=shape, actions=nA, device="cuda", architecture="single", hidden_units=256,
                        head_scale=0.1, head_bias=True)
r = rollout.Runner(model, logger.Logger(quiet=True))
r.vec_env = envs.create_envs_classic()
r.reset()
r.generate_rollout()
r.calculate_returns()
spec = os.environ.get("PPO_AB", "models.ADAM_SCATTER=0,1")
target, values = spec.split("=")
mod_name, attr = target.split(".")
mod = {"models": models, "rollout": rollout}[mod_name]
variants = {f"{target}={v}": int(v) for v in values.split(",")}
times = {k: [] for k in variants}
n_mb = 2 * (r.N * r.A // 256)
for rnd in range(rounds + 1):
    for name, v in variants.items():
        setattr(mod, attr, v)
        r.train()  # untimed: scratch buffers / pointer tables of this variant
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.train()
        torch.cuda.synchronize()
        if rnd:
            times[name].append((time.perf_counter() - t0) / n_mb * 1e3)
for name, t in times.items():
    print(f"{name:40s} median {np.median(t):.4f}  mean {np.mean(t):.4f}  min {np.min(t):.4f} ms per minibatch  ({len(t)} x train())")
