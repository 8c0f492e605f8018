"""GPU: the Runner's data-parallel path with two ranks sharing the one GPU of the test box (gloo carries the
CUDA tensors; on a node it is nccl = RCCL, one rank per GPU — same code path above the backend).

What must hold: ranks own different env columns, all-reduce the flat gradient once per
optimiser step and the advantage moments once per batch, and therefore hold IDENTICAL parameters and Adam
state after every step; the normalised advantages of both ranks use the GLOBAL mean / std."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, hashlib
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from ppo_amd import envs, logger, models, parallel, rollout
from ppo_amd.config import args
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
args.setup(["--agents=32", "--n_steps=8", "--model_architecture=single", "--model_encoder=impala",
            "--env_type=synthetic", "--env_embed_time=False", "--seed=3", "--device=cuda",
            "--policy_opt_mini_batch_size=128", "--policy_opt_epochs=2", "--disable_logging=True"])
torch.manual_seed(1000 + rank)     # DIFFERENT initial weights per rank (what --seed=-1 gives): the Runner must
np.random.seed(100 + rank)         # broadcast rank 0's replica; different minibatch permutations
shape, nA = envs.get_env_spec()
model = models.TVFModel("impala", input_dims=shape, actions=nA, device="cuda", architecture="single",
                        hidden_units=256, head_scale=0.1, head_bias=True)
before = hashlib.sha256(model.policy_net.flat.cpu().numpy().tobytes()).hexdigest()
r = rollout.Runner(model, logger.Logger(quiet=True))
assert r.world == 2 and r.rank == rank
after = hashlib.sha256(model.policy_net.flat.cpu().numpy().tobytes()).hexdigest()
both = [None, None]
dist.all_gather_object(both, (before, after))
assert both[0][0] != both[1][0], "the test did not start the replicas apart"
assert both[0][1] == both[1][1] == both[0][0], "Runner.__init__ did not broadcast rank 0's parameters"
assert r._reducers and model.policy_net.grad_ready_hook is not None, "bucketed gradient reducer not installed"
r.vec_env = envs.create_envs_classic(rank=rank, world=world)
r.reset()
for it in range(2):
    r.generate_rollout()
    r.calculate_returns()
    r.train()
torch.cuda.synchronize()
assert r.step == 2 * 8 * 32 * 2                         # env steps counted over both ranks
assert r.net._adam_step == 2 * 2 * (8 * 32 // 64)       # local minibatch = 128 / 2
# identical parameters and optimiser state on both ranks
digest = hashlib.sha256(r.net.flat.cpu().numpy().tobytes() + r.net.exp_avg.cpu().numpy().tobytes()).hexdigest()
obs_digest = hashlib.sha256(r.all_obs[0].cpu().numpy().tobytes()).hexdigest()
gathered = [None, None]
dist.all_gather_object(gathered, (digest, obs_digest, float(r._mean_std[0]), float(r._mean_std[1])))
assert gathered[0][0] == gathered[1][0], "parameters diverged across ranks"
assert gathered[0][1] != gathered[1][1], "both ranks stepped the same envs"
assert gathered[0][2:] == gathered[1][2:], "advantage moments were not reduced over ranks"
# the global moments really are the moments of both ranks' advantages together
adv = r.advantage.double().flatten()
m = torch.stack([adv.sum(), (adv * adv).sum(), torch.tensor(float(adv.numel()), dtype=torch.float64, device=adv.device)])
parallel.allreduce_sum_(m)
mean, var = parallel.mean_var_from_moments(m.cpu())
assert abs(mean - float(r._mean_std[0])) < 1e-5 * max(1.0, abs(mean)) and abs(var ** 0.5 - float(r._mean_std[1])) < 1e-5 * max(1.0, var ** 0.5)
assert torch.isfinite(r.net.flat).all()
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_rank_runner_keeps_replicas_identical(tmp_path):
    script = tmp_path / "dp_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
        assert "ok" in o


CONFIG4_WORKER = r'''
import os, sys, hashlib
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from ppo_amd import envs, logger, models, parallel, rollout
from ppo_amd.config import args
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
# BASELINE.json configs[3] as one rank sees it: Breakout-shaped (4x84x84 uint8, 4 actions), 512 envs per rank,
# the GLOBAL minibatch split over ranks (256 per rank here), short rollout to keep the test quick
args.setup(["--agents=512", "--n_steps=4", "--model_architecture=single", "--model_encoder=impala",
            "--env_type=synthetic", "--env_synthetic_actions=4", "--env_embed_time=False", "--seed=3", "--device=cuda",
            "--policy_opt_mini_batch_size=512", "--policy_opt_epochs=2", "--disable_logging=True"])
np.random.seed(100 + rank)
shape, nA = envs.get_env_spec()
assert shape == (4, 84, 84) and nA == 4
model = models.TVFModel("impala", input_dims=shape, actions=nA, device="cuda", architecture="single",
                        hidden_units=256, head_scale=0.1, head_bias=True)
r = rollout.Runner(model, logger.Logger(quiet=True))
r.vec_env = envs.create_envs_classic(rank=rank, world=world)
r.reset()
for it in range(2):
    r.generate_rollout()
    r.calculate_returns()
    r.train()
torch.cuda.synchronize()
assert r.step == 2 * 4 * 512 * 2
assert r.net._adam_step == 2 * 2 * (4 * 512 // 256)          # local minibatch = 512 / 2
assert int(r.actions.max()) <= 3 and int(r.actions.min()) >= 0
digest = parallel.assert_identical_across_ranks([r.net.flat, r.net.exp_avg, r.net.exp_avg_sq], "replicas after 2 iterations")
obs_digest = hashlib.sha256(r.all_obs[0, :8].cpu().numpy().tobytes()).hexdigest()
got = [None, None]
dist.all_gather_object(got, obs_digest)
assert got[0] != got[1], "both ranks stepped the same envs"
ms = r._reducers[id(r.net)].exposed_ms()
assert ms is not None and ms >= 0.0
assert torch.isfinite(r.net.flat).all()
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok", digest[:12])
'''


def test_config4_shard_two_ranks(tmp_path):
    """BASELINE.json configs[3] (Breakout, 4096 envs over 8 GPUs) at the size one rank sees: A = 512, 4 actions,
    global-minibatch semantics, bucketed + overlapped gradient reduction; replicas bit-identical afterwards."""
    script = tmp_path / "c4_worker.py"
    script.write_text(CONFIG4_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29549", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    try:
        outs = [p.communicate(timeout=420)[0] for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
        assert "ok" in o
