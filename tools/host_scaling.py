"""Host-contention PROXY for the 8-ranks-on-one-node case, runnable with ONE GPU (not a scaling curve).

A rank's rollout has a host leg on its critical path: 16 env threads generate a group's observations (~0.1 ms per group
step) and the caller queues the pinned uploads.  With 8 ranks on a node those legs compete for cores, memory bandwidth
and (across sockets) for the fabric.  What one GPU can show of that: ONE real rank (this process: the bench
configuration's Runner, rollouts only) beside L GPU-FREE sibling processes that run the same synthetic env's step loop
with the same thread count, paced at a real rank's cadence (two groups of 128 envs, one group step every `--period-ms`).
Siblings are fresh children that never touch HIP (pageable buffers, no torch.cuda call); each takes the CPU share
ppo_amd/affinity.py would give local rank k of L + 1 unless --no-affinity.

    python tools/host_scaling.py [--levels 0,1,3,7] [--n-steps 64] [--rollouts 6] [--no-affinity] [--period-ms 0.23]

Prints one JSON line per level and a markdown table (commit it under profiles/ labelled as a proxy).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sibling(a):
    """GPU-free: the env step loop of one rank (two groups, 16 threads), paced; runs until the parent closes stdin."""
    from ppo_amd import affinity
    allowed = [int(x) for x in os.environ["HOST_SCALING_ALLOWED"].split(",")]
    os.sched_setaffinity(0, allowed)  # the parent's own narrowed mask was inherited: start from the whole allowance
    if not a.no_affinity:
        cpus = affinity.plan(a.index, a.of, allowed)
        if cpus:
            os.sched_setaffinity(0, cpus)
    import numpy as np
    from ppo_amd.vec_env import SyntheticVecEnv
    groups = [SyntheticVecEnv(128, seed=100 + a.index, env_offset=i * 128, threads=a.threads, pinned=False) for i in range(2)]
    for g in groups:
        g.reset()
    act = np.zeros(128, np.int32)
    period = a.period_ms * 1e-3
    print("ready", flush=True)
    import select
    steps, t_next = 0, time.perf_counter()
    while True:
        for g in groups:
            g.step_arrays(act)
            steps += 1
            t_next += period
            while time.perf_counter() < t_next:  # a rank's host thread waits on a device event here; spin = worst case
                pass
        if steps % 512 == 0 and select.select([sys.stdin], [], [], 0)[0]:
            return


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--levels", default="0,1,3,7")
    p.add_argument("--n-steps", type=int, default=64)
    p.add_argument("--rollouts", type=int, default=6)
    p.add_argument("--threads", type=int, default=16)
    p.add_argument("--period-ms", type=float, default=0.23)
    p.add_argument("--no-affinity", action="store_true")
    p.add_argument("--sibling", action="store_true")
    p.add_argument("--index", type=int, default=0)
    p.add_argument("--of", type=int, default=1)
    a = p.parse_args()
    if a.sibling:
        return sibling(a)
    levels = [int(x) for x in a.levels.split(",")]
    from ppo_amd import affinity
    allowed = sorted(os.sched_getaffinity(0))

    import numpy as np
    import torch
    from ppo_amd import envs, logger, models, rollout
    from ppo_amd.config import args
    N = a.n_steps
    args.setup(["--agents=256", f"--n_steps={N}", "--model_architecture=single", "--model_encoder=impala",
                "--env_type=synthetic", "--env_embed_time=False", "--seed=1", "--device=cuda",
                "--policy_opt_mini_batch_size=256", "--disable_logging=True", "--upload_batch=True",
                "--env_reward_normalization=off", f"--env_synthetic_threads={a.threads}"])
    rows = []
    for L in levels:
        # this rank's own share for an (L + 1)-rank node, set BEFORE its env threads exist
        os.sched_setaffinity(0, allowed)
        mine = None if a.no_affinity else affinity.plan(0, L + 1, allowed)
        if mine:
            os.sched_setaffinity(0, mine)
        torch.manual_seed(1)
        np.random.seed(1)
        shape, nA = envs.get_env_spec()
        model = models.TVFModel("impala", input_dims=shape, actions=nA, device="cuda", architecture="single",
                                hidden_units=256, head_scale=0.1, head_bias=True)
        r = rollout.Runner(model, logger.Logger(quiet=True))
        r.vec_env = envs.create_envs_classic()
        r.reset()
        r.generate_rollout()
        torch.cuda.synchronize()
        sibs = []
        try:
            for k in range(1, L + 1):
                cmd = [sys.executable, os.path.abspath(__file__), "--sibling", "--index", str(k), "--of", str(L + 1),
                       "--threads", str(a.threads), "--period-ms", str(a.period_ms)] + (["--no-affinity"] if a.no_affinity else [])
                # a fresh child (it sets its own mask): an ordinary child process, never an exec of this GPU-holding one
                sibs.append(subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True,
                                             env=dict(os.environ, HOST_SCALING_ALLOWED=",".join(map(str, allowed)))))
            for s in sibs:
                assert s.stdout.readline().strip() == "ready"
            time.sleep(0.2)
            times = []
            for _ in range(a.rollouts):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r.generate_rollout()
                torch.cuda.synchronize()
                times.append((time.perf_counter() - t0) / (N + 1) * 1e3)
        finally:
            for s in sibs:
                try:
                    s.stdin.write("stop\n")
                    s.stdin.flush()
                except OSError:
                    pass
            for s in sibs:
                try:
                    s.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    s.kill()  # exactly the child started above
            r.vec_env.close()
        row = {"siblings": L, "ranks_emulated": L + 1, "ms_per_env_step_median": round(float(np.median(times)), 4),
               "min": round(min(times), 4), "max": round(max(times), 4), "cpus_of_this_rank": len(mine) if mine else len(allowed),
               "affinity": not a.no_affinity}
        rows.append(row)
        print(json.dumps(row), flush=True)
        del r, model
    base = rows[0]["ms_per_env_step_median"]
    print(f"\n| ranks emulated (1 real + GPU-free siblings) | cores of the real rank | rollout ms per env step (median of {a.rollouts}) | min .. max | vs alone |")
    print("|---|---|---|---|---|")
    for row in rows:
        print(f"| {row['ranks_emulated']} | {row['cpus_of_this_rank']} | {row['ms_per_env_step_median']:.4f} | "
              f"{row['min']:.4f} .. {row['max']:.4f} | {row['ms_per_env_step_median'] / base:.3f} |")


if __name__ == "__main__":
    main()
