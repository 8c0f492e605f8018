"""Interleaved A/B of rollout switches inside ONE process (run-to-run variation between processes on a GPU box is several
per cent, more than most of these switches move): variants alternate rollout by rollout on the same Runner.
Usage: [PPO_EXTRA_ARGS="--agents=1024 ..."] python tools/rollout_ab.py [n_steps] [rounds]
variants: models.FUSE_BLOCK, models.CHAIN_SPLIT, rollout.FUSE_ACT; PPO_AB=conv1: the first layer's two kernels
(ppo_conv1_pool_form: LDS form / pooled out of the accumulators), everything else on"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_amd import envs, logger, models, rollout  # noqa: E402
from ppo_amd.config import args  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
args.setup(["--agents=256", f"--n_steps={N}", "--model_architecture=single", "--model_encoder=impala", "--env_type=synthetic",
            "--env_embed_time=False", "--seed=1", "--device=cuda", "--policy_opt_mini_batch_size=256", "--policy_opt_epochs=2",
            "--disable_logging=True", "--upload_batch=True", "--env_reward_normalization=off"]
           + os.environ.get("PPO_EXTRA_ARGS", "").split())  # e.g. the procgen shape: --agents=1024 --env_synthetic_shape=3,64,64 ...
torch.manual_seed(1)
np.random.seed(1)
shape, nA = envs.get_env_spec()
model = models.TVFModel("impala", input_dims=shape, actions=nA, device="cuda", architecture="single", hidden_units=256,
                        head_scale=0.1, head_bias=True)
r = rollout.Runner(model, logger.Logger(quiet=True))
r.vec_env = envs.create_envs_classic()
r.reset()
variants = {"base": (0, 0, 0), "block": (1, 0, 0), "block+act": (1, 0, 1), "block+split": (1, 1, 0), "block+split+act": (1, 1, 1)}
if os.environ.get("PPO_AB") == "chunks":  # pieces a group's observations go up in (all kernel switches on)
    variants = {f"upload chunks {c}": (1, 1, 1, c) for c in (1, 2, 4, 8)}
if os.environ.get("PPO_AB") == "graph":  # hipGraph of a group's forward (fixed staging buffer, separate sampling launch)
    variants = {"eager": (1, 1, 1, 2, 0), "graph": (1, 1, 1, 2, 1)}
conv1_form = None
if os.environ.get("PPO_AB") == "conv1":
    from ppo_amd import _lib
    conv1_form = {"first layer: LDS form": 1, "first layer: from the accumulators": 0}
    variants = {k: (1, 1, 1) for k in conv1_form}
times = {k: [] for k in variants}
for rnd in range(rounds + 1):
    for name, (blk, split, act, *rest) in variants.items():
        models.FUSE_BLOCK, models.CHAIN_SPLIT, rollout.FUSE_ACT = blk, split, act
        if conv1_form:
            _lib.load().ppo_conv1_pool_form(conv1_form[name])
        if rest:
            rollout.UPLOAD_CHUNKS = rest[0]
        if len(rest) > 1:
            rollout.ROLLOUT_GRAPH = rest[1]
            r._graphs.clear()
        model.policy_net._plans.clear()
        r.generate_rollout()  # records the launch lists of this variant
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.generate_rollout()
        torch.cuda.synchronize()
        if rnd:
            times[name].append((time.perf_counter() - t0) / (N + 1) * 1e3)
for name, t in times.items():
    print(f"{name:18s} median {np.median(t):.4f}  mean {np.mean(t):.4f}  min {np.min(t):.4f} ms per env step  ({len(t)} rollouts)")
