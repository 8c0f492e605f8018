"""CPU: host-side logic — the synthetic vector env (C++ threads in libppo_amd.so, no GPU needed),
the rl.config mirror, the drop-in `rl.*` module names, and the data-parallel helpers under a
2-process gloo group (the N>1 path's sharding and reductions)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_synthetic_env_is_deterministic_and_thread_count_independent(hip_lib):
    from ppo_amd.vec_env import SyntheticVecEnv
    A, shape = 24, (4, 84, 84)
    runs = []
    for threads in (0, 1, 5):
        env = SyntheticVecEnv(A, shape, 6, seed=7, p_done=0.2, threads=threads, pinned=False)
        obs0 = env.reset().copy()
        traj = [obs0]
        rng = np.random.default_rng(0)
        for _ in range(6):
            obs, rew, done, infos = env.step(rng.integers(0, 6, A))
            traj += [obs.copy(), rew, done, np.asarray([i["time"] for i in infos])]
        runs.append(traj)
        env.close()
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert np.array_equal(a, b)
    obs = runs[0][1]
    assert obs.dtype == np.uint8 and obs.shape == (A, *shape)
    assert 120 < obs.mean() < 135 and obs.std() > 70  # uniform[0,255]
    assert not np.array_equal(runs[0][0], runs[0][1])


def test_synthetic_env_statistics_skip_and_autoreset(hip_lib):
    from ppo_amd.vec_env import SyntheticVecEnv
    A = 512
    env = SyntheticVecEnv(A, (2, 8, 8), 4, seed=3, p_done=0.05, threads=2, pinned=False)
    env.reset()
    rews, dones = [], []
    for t in range(200):
        _, r, d, infos = env.step(np.zeros(A, np.int32))
        rews.append(r)
        dones.append(d)
    rews, dones = np.stack(rews), np.stack(dones)
    assert abs(rews.mean()) < 0.02 and abs(rews.std() - 1) < 0.02       # N(0,1)
    assert abs(dones.mean() - 0.05) < 0.005                               # Bernoulli(p)
    # time counts steps since the last reset; a done step reports the finished episode's length
    t_now = np.asarray([i["time"] for i in infos])
    last_done = np.where(dones.any(0), 199 - np.argmax(dones[::-1], 0), -1)
    ongoing = ~dones[-1]
    assert np.array_equal(t_now[ongoing], (199 - last_done)[ongoing])
    # action -1 leaves an env untouched
    before = env.obs.copy()
    a = np.zeros(A, np.int32)
    a[::2] = -1
    obs, r, d, _ = env.step(a)
    assert np.array_equal(obs[::2], before[::2]) and not np.array_equal(obs[1::2], before[1::2])
    assert (r[::2] == 0).all() and not d[::2].any()


def test_synthetic_env_sharding_equals_one_big_env(hip_lib):
    """Every value depends on the GLOBAL env index: two shards of 8 envs == envs 0..15 of one env."""
    from ppo_amd.vec_env import SyntheticVecEnv
    big = SyntheticVecEnv(16, (1, 4, 4), 3, seed=11, p_done=0.1, threads=0, pinned=False)
    lo = SyntheticVecEnv(8, (1, 4, 4), 3, seed=11, p_done=0.1, env_offset=0, threads=0, pinned=False)
    hi = SyntheticVecEnv(8, (1, 4, 4), 3, seed=11, p_done=0.1, env_offset=8, threads=0, pinned=False)
    assert np.array_equal(big.reset(), np.concatenate([lo.reset(), hi.reset()]))
    for _ in range(5):
        ob, rb, db, _ = big.step(np.ones(16, np.int32))
        ol, rl_, dl, _ = lo.step(np.ones(8, np.int32))
        oh, rh, dh, _ = hi.step(np.ones(8, np.int32))
        assert np.array_equal(ob, np.concatenate([ol, oh]))
        assert np.array_equal(rb, np.concatenate([rl_, rh])) and np.array_equal(db, np.concatenate([dl, dh]))


def test_config_mirrors_reference_flags_and_defaults():
    from ppo_amd.config import Config
    c = Config().setup([])
    # rl/config.py defaults
    assert (c.agents, c.n_steps, c.gamma, c.lambda_policy, c.lambda_value) == (256, 256, 0.999, 0.95, 0.95)
    assert (c.ppo_epsilon, c.entropy_bonus, c.ppo_vf_coef, c.max_grad_norm) == (0.2, 0.01, 0.5, 20.0)
    assert (c.policy_opt.lr, c.policy_opt.adam_epsilon, c.policy_opt.epochs, c.policy_opt.mini_batch_size) == (2.5e-4, 1e-5, 2, 256)
    assert c.model.head_scale == 0.1 and c.model.head_bias is True and c.model.architecture == "dual"
    assert c.batch_size == 65536 and c.tvf_return_n_step == 20
    c = Config().setup(["--agents=8", "--model_architecture=single", "--policy_opt_lr=1e-3", "--env_embed_time=False",
                        "--upload_batch", "--replay_size=5"])
    assert c.agents == 8 and c.policy_opt.lr == 1e-3 and c.env.embed_time is False and c.upload_batch is True
    assert c._ignored == ["--replay_size=5"]
    with pytest.raises(ValueError):
        Config().setup(["--model_architecture=triple"])
    with pytest.raises(ValueError):
        Config().setup(["--grad_clip_mode=banana"])


def test_drop_in_module_names_resolve():
    import rl.config
    import rl.logger
    import rl.returns
    from ppo_amd import returns
    assert rl.returns.gae is returns.gae and rl.returns.td_lambda is returns.td_lambda
    assert rl.returns.calculate_bootstrapped_returns is returns.calculate_bootstrapped_returns
    assert hasattr(rl.config, "args") and hasattr(rl.logger, "Logger")
    src = open(os.path.join(ROOT, "rl", "rollout.py")).read() + open(os.path.join(ROOT, "rl", "ppo.py")).read()
    assert "ppo_amd.rollout" in src and "ppo_amd.ppo" in src


def test_product_code_never_imports_the_oracle():
    for d, _, files in os.walk(os.path.join(ROOT, "ppo_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(d, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
    for f in ("train.py",):
        assert "oracle" not in open(os.path.join(ROOT, f)).read()


GLOO_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from ppo_amd import parallel
from ppo_amd.vec_env import SyntheticVecEnv
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
w, r = parallel.world_size(), parallel.rank()
assert w == 2
# env sharding: this rank's block of the global env set
off, n = parallel.shard(6)
assert off == r * 6 and n == 6
env = SyntheticVecEnv(n, (1, 4, 4), 3, seed=5, p_done=0.1, env_offset=off, threads=0, pinned=False)
big = SyntheticVecEnv(12, (1, 4, 4), 3, seed=5, p_done=0.1, threads=0, pinned=False)
assert np.array_equal(env.reset(), big.reset()[off:off + n])
o, rew, d, _ = env.step(np.zeros(n, np.int32)); ob, rb, db, _ = big.step(np.zeros(12, np.int32))
assert np.array_equal(o, ob[off:off + n]) and np.array_equal(rew, rb[off:off + n])
# advantage moments: all-reduced {sum, sumsq, n} give the GLOBAL mean / variance
rng = np.random.default_rng(0); full = rng.normal(2.0, 3.0, size=(16, 12)).astype(np.float32)
mine = full[:, off:off + n].astype(np.float64)
m = torch.tensor([mine.sum(), (mine ** 2).sum(), mine.size], dtype=torch.float64)
parallel.allreduce_sum_(m)
mean, var = parallel.mean_var_from_moments(m)
assert abs(mean - full.astype(np.float64).mean()) < 1e-12 and abs(var - full.astype(np.float64).var()) < 1e-10
# gradient exchange: sum over ranks, divided by world inside the optimiser == mean gradient
g = torch.full((1000,), float(r + 1)); parallel.allreduce_sum_(g)
assert torch.equal(g / w, torch.full((1000,), 1.5))
# the global minibatch flag is split across ranks
assert parallel.local_minibatch(256) == 128
try:
    parallel.local_minibatch(255); raise SystemExit("expected ValueError")
except ValueError:
    pass
dist.barrier(); dist.destroy_process_group()
print("rank", r, "ok")
'''


def test_data_parallel_helpers_two_ranks_gloo(hip_lib, tmp_path):
    script = tmp_path / "gloo_worker.py"
    script.write_text(GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o


def test_annealing_follows_the_reference_schedule():
    """Runner.anneal / learning-rate properties (rl/rollout.py:331-392) — pure host arithmetic, checked on a
    stand-in object so no GPU is needed."""
    import math
    import types

    from ppo_amd import rollout
    from ppo_amd.config import args
    args.setup(["--epochs=10", "--policy_opt_lr=0.001", "--policy_opt_lr_anneal=True", "--ppo_epsilon_anneal=True"])
    r = types.SimpleNamespace(step=2.5e6)
    anneal = lambda x, mode="linear": rollout.Runner.anneal(r, x, mode)  # noqa: E731
    r.anneal = anneal
    assert anneal(2.0, "off") == 2.0
    assert abs(anneal(2.0, "linear") - 1.5) < 1e-12 and abs(anneal(2.0, "linear_inc") - 0.5) < 1e-12
    assert abs(anneal(2.0, "quad_inc") - 2.0 * 0.0625) < 1e-12
    assert abs(anneal(1.0, "cos") - (1 + math.cos(math.pi * 2 * 2.5e6 / 20e6)) / 2) < 1e-12
    assert abs(rollout.Runner._lr(r, args.policy_opt) - 0.00075) < 1e-12
    assert rollout.Runner._lr(r, args.value_opt) == args.value_opt.lr  # not annealed
    r.step = 20e6  # past the end: clipped at 0
    assert anneal(2.0, "linear") == 0.0 and anneal(2.0, "linear_inc") == 2.0
    args.setup(["--epochs=10", "--anneal_target_epoch=5"])
    r.step = 2.5e6
    assert abs(anneal(2.0, "linear") - 1.0) < 1e-12
    args.setup([])


DESYNC_WORKER = r'''
import os, sys, types
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from ppo_amd import envs, parallel, ppo


def main():
    from ppo_amd.config import args
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args.setup(["--agents=4", "--env_type=classic", "--env_name=CartPole", "--env_reward_normalization=rms",
                "--seed=5", "--workers=2"])
    np.random.seed(11 + 97 * rank)          # ranks draw DIFFERENT warm-up lengths
    vec = envs.create_envs_classic(rank=rank, world=world)
    calls = [0]
    from ppo_amd import wrappers
    norm = wrappers.get_wrapper(vec, wrappers.VecNormalizeRewardWrapper)
    inner = norm.moments_sync
    assert inner is not None, "data-parallel reward normalisation must reduce its moments over ranks"
    def counted(m):
        calls[0] += 1
        return inner(m)
    norm.moments_sync = counted
    runner = types.SimpleNamespace(A=4, world=world, n_actions=2, vec_env=vec, obs=vec.reset(), device="cpu",
                                   model=types.SimpleNamespace(obs_norm=None))
    ppo.desync_envs(runner, 1, 9)
    got = [None] * world
    dist.all_gather_object(got, calls[0])
    assert got[0] == got[1] == 9, got       # every rank issued the same number of collectives: max_duration
    # and the collective that follows pairs up (a mismatch would hang or mix a moments reduce into it)
    g = torch.full((1000,), float(rank + 1)); parallel.allreduce_sum_(g)
    assert torch.equal(g, torch.full((1000,), 3.0))
    # both ranks hold the same reward statistics, fed by all 8 envs
    stats = [None] * world
    dist.all_gather_object(stats, (float(norm.ret_rms.mean), float(norm.ret_rms.var), float(norm.ret_rms.count)))
    assert stats[0] == stats[1], stats
    vec.close()
    dist.barrier(); dist.destroy_process_group()
    print("rank", rank, "ok")


if __name__ == "__main__":  # the env pool spawns workers, which re-import this file
    main()
'''


def test_desync_envs_issues_the_same_collectives_on_every_rank(hip_lib, tmp_path):
    """with gym-API envs and rms reward normalisation every env step all-reduces three moments; the
    warm-up length is drawn per rank, so data-parallel runs must not let it set the loop length."""
    script = tmp_path / "desync_worker.py"
    script.write_text(DESYNC_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    try:
        outs = [p.communicate(timeout=180)[0] for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
        assert "ok" in o


def _bench(*argv, env=None, timeout=180):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus N` is the driver's single-command form: it must start N ranks itself (gloo dry run
    here: rendezvous + one all-reduce, no GPU), and must never fall back to a silent N = 1."""
    import json
    r = _bench("--gpus", "2", "--rendezvous-only", "--backend", "gloo")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["ranks_seen"] == 2 and line["n_gpus"] == 2 and line["backend"] == "gloo"
    # more GPUs than the node has: refused before any rank starts (this container has none)
    r = _bench("--gpus", "2")
    assert r.returncode != 0 and "exposes" in (r.stderr + r.stdout)
    # a launcher that set WORLD_SIZE to something else: refused, not reported as --gpus
    r = _bench("--gpus", "2", "--rendezvous-only", "--backend", "gloo", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
    # N = 1 takes no launcher and no process group
    r = _bench("--gpus", "1", "--rendezvous-only")
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["ranks_seen"] == 1


def test_bench_launcher_stops_everything_when_a_rank_dies_or_hangs():
    """One rank exits before the rendezvous: the launcher must notice (it polls every rank, not just rank 0), stop
    the rank still waiting in init_process_group, show the dead rank's stderr and exit non-zero well inside the
    process group's own timeout.  A rank that never joins is ended by the wall limit the same way."""
    import time
    t0 = time.time()
    r = _bench("--gpus", "2", "--rendezvous-only", "--backend", "gloo", "--fail-rank", "1", "--dist-timeout", "120")
    assert r.returncode != 0
    assert "rank(s) failed" in r.stderr and "(1, 1)" in r.stderr and "told to fail" in r.stderr, r.stderr[-2000:]
    assert time.time() - t0 < 60, "the launcher waited for the collective timeout instead of the dead rank"
    t0 = time.time()
    r = _bench("--gpus", "2", "--rendezvous-only", "--backend", "gloo", "--hang-rank", "0", "--wall-limit", "8",
               "--dist-timeout", "120")
    assert r.returncode != 0 and "wall limit" in r.stderr, r.stderr[-2000:]
    assert time.time() - t0 < 60


def test_u8_unit_recipe_is_the_exact_quotient():
    """csrc/conv_stage.h u8_unit: q = x * fl(1/255); q = fma(fma(q, -255, x), fl(1/255), q) equals fl(x / 255) for every
    uint8 x.  float32 FMAs are emulated exactly in float64 (products and sums of these magnitudes are exact there)."""
    x = np.arange(256, dtype=np.float32)
    want = x / np.float32(255.0)
    r = np.float32(1.0) / np.float32(255.0)

    def fma(a, b, c):
        return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(np.float32)

    q = (x * r).astype(np.float32)
    assert int((q != want).sum()) > 0, "the bare reciprocal product is NOT exact: the correction is needed"
    got = fma(fma(q, np.float32(-255.0), x), r, q)
    assert np.array_equal(got, want)


def test_affinity_plan_on_a_fake_two_socket_node(tmp_path):
    """ppo_amd/affinity.py: GPU -> NUMA node from the KFD topology and the PCI device's numa_node, the node's cores
    divided among the ranks whose GPUs share it, restricted to what the process may use; fallbacks without NUMA data."""
    from ppo_amd import affinity
    assert affinity.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    root = tmp_path
    nodes = root / "sys/class/kfd/kfd/topology/nodes"
    # KFD nodes 0, 1 are the CPU sockets (simd_count 0); GPUs 2..9: four per socket, render minors 128..135
    for i in range(10):
        d = nodes / str(i)
        d.mkdir(parents=True)
        gpu = i >= 2
        (d / "properties").write_text(f"cpu_cores_count {0 if gpu else 64}\nsimd_count {1024 if gpu else 0}\n"
                                      + (f"drm_render_minor {126 + i}\n" if gpu else ""))
        if gpu:
            dev = root / f"sys/class/drm/renderD{126 + i}/device"
            dev.mkdir(parents=True)
            (dev / "numa_node").write_text(f"{0 if i < 6 else 1}\n")
    for n, cpus in ((0, "0-63,128-191"), (1, "64-127,192-255")):
        d = root / f"sys/devices/system/node/node{n}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")
    r = str(root)
    assert affinity.gpu_numa_nodes(r) == [0, 0, 0, 0, 1, 1, 1, 1]
    allowed = range(256)
    plans = [affinity.plan(k, 8, allowed, r, env={}) for k in range(8)]
    assert all(len(p) == 32 for p in plans)
    node0 = set(affinity.parse_cpulist("0-63,128-191"))
    assert all(set(p) <= node0 for p in plans[:4]) and all(not (set(p) & node0) for p in plans[4:])
    assert len(set().union(*map(set, plans))) == 256  # disjoint shares that cover the machine
    # one rank alone on its socket gets the whole socket; a cpuset restriction is respected
    assert set(affinity.plan(0, 1, allowed, r, env={})) == node0
    assert affinity.plan(5, 8, range(64, 96), r, env={}) == list(range(72, 80))  # node 1 allowed cores 64..95, rank 5 = 2nd of 4
    # HIP_VISIBLE_DEVICES reorders which GPU a local rank drives
    assert not (set(affinity.plan(0, 2, allowed, r, env={"HIP_VISIBLE_DEVICES": "4,0"})) & node0)
    # no NUMA information (numa_node -1) or no topology at all: an even split of what is allowed
    for i in range(2, 10):
        (root / f"sys/class/drm/renderD{126 + i}/device/numa_node").write_text("-1\n")
    assert affinity.plan(1, 4, range(16), r, env={}) == [4, 5, 6, 7]
    assert affinity.plan(1, 2, range(8), str(tmp_path / "nothing"), env={}) == [4, 5, 6, 7]
    assert affinity.plan(0, 1, [3], r, env={}) is None


def test_affinity_release_hands_every_thread_its_mask_back():
    """What bench.py does around its CPU-baseline leg: threads started while the rank was pinned inherited the narrow
    mask; release() widens all of them again (in a child process, so that the test runner's own mask is not touched)."""
    import subprocess
    import sys
    code = r'''
import os, threading, time
from ppo_amd import affinity
before = os.sched_getaffinity(0)
if len(before) < 2:
    print("SKIP"); raise SystemExit(0)
assert affinity.release() is False          # nothing was pinned yet
narrow = sorted(before)[:1]
os.sched_setaffinity(0, narrow)
affinity._unpinned_mask = set(before)        # (what pin_rank records when it narrows the mask)
stop = threading.Event()
seen = {}
def worker():
    seen["start"] = os.sched_getaffinity(0)
    stop.wait()
    seen["end"] = os.sched_getaffinity(0)
t = threading.Thread(target=worker); t.start()
time.sleep(0.05)
assert affinity.release() is True
stop.set(); t.join()
assert seen["start"] == set(narrow) and seen["end"] == before and os.sched_getaffinity(0) == before
print("OK")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() in ("OK", "SKIP")
