"""Where does a rollout step go?  Host enqueue time vs GPU time of the policy step, env step time and the
H2D copy, for the bench configuration.  Usage: python tools/rollout_profile.py [agents] [parts]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_amd import envs, logger, models, rollout  # noqa: E402
from ppo_amd.config import args  # noqa: E402

A = int(sys.argv[1]) if len(sys.argv) > 1 else 256
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
args.setup([f"--agents={A}", "--n_steps=64", "--model_architecture=single", "--model_encoder=impala",
            "--env_type=synthetic", "--env_embed_time=False", "--seed=1", "--device=cuda", "--disable_logging=True",
            f"--env_pipeline_parts={parts}"])
shape, nA = envs.get_env_spec()
model = models.TVFModel("impala", input_dims=shape, actions=nA, device="cuda", architecture="single",
                        hidden_units=256, head_scale=0.1, head_bias=True)
r = rollout.Runner(model, logger.Logger(quiet=True))
r.vec_env = envs.create_envs_classic()
r.reset()
r.generate_rollout()
torch.cuda.synchronize()

t0 = time.perf_counter()
for _ in range(3):
    r.generate_rollout()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"rollout: {dt * 1e3:.1f} ms for {r.N + 1} steps -> {dt / (r.N + 1) * 1e3:.3f} ms/step, "
      f"{r.N * A / dt:.0f} env-steps/s (rollout only)")

env_parts = getattr(r.vec_env, "parts", [r.vec_env])
per = A // len(env_parts)
reps = 50
# host enqueue time and GPU time of one group's policy step
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for _ in range(reps):
    r._policy_step(3, 0, per)
e1.record()
host = (time.perf_counter() - t0) / reps
torch.cuda.synchronize()
print(f"policy step B={per}: host enqueue {host * 1e3:.3f} ms, GPU {e0.elapsed_time(e1) / reps:.3f} ms")
# env step
act = np.zeros(per, np.int32)
t0 = time.perf_counter()
for _ in range(reps):
    env_parts[0].step_arrays(act)
print(f"env step of {per} envs: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms")
# H2D
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    r.all_obs[3, :per].copy_(env_parts[0].obs_t, non_blocking=True)
torch.cuda.synchronize()
print(f"H2D of {per} obs: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms")
# D2H + sync latency
t0 = time.perf_counter()
for _ in range(reps):
    r._actions_host[:per].copy_(r.actions[3, :per], non_blocking=True)
    torch.cuda.current_stream().synchronize()
print(f"D2H actions + sync: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms")
