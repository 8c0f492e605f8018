"""Training loop — host mirror of the reference's rl/ppo.py (`train(model, log)` :47-383,
`desync_envs` :21-44).  One iteration = generate_rollout -> calculate_returns -> train, each timed
like the reference (`time_rollout`, `time_returns`, `time_train`, :256-284); throughput is the
reference's IPS = env steps / wall seconds (:354-365), summed over ranks.
"""
import math
import os
import time

import numpy as np
import torch

from . import envs
from .config import args
from .logger import Logger, LogVariable
from .rollout import Runner


def desync_envs(runner, min_duration: int, max_duration: int, verbose=True):
    """Run every env for a random number of steps with random actions so episodes are out of phase
    (rl/ppo.py:21-44); envs that are done warming up receive action -1 (skip)."""
    if max_duration <= 0:
        return
    A = runner.A
    steps = np.random.randint(min_duration, max_duration + 1, size=A)
    norm = getattr(runner.model, "obs_norm", None)
    # with observation normalisation every warm-up step also feeds the running statistics (rl/ppo.py:31), and
    # data-parallel ranks reduce them together, so all ranks run the same number of steps
    n_iter = int(max_duration) if norm is not None and runner.world > 1 else int(steps.max())
    for t in range(n_iter):
        if norm is not None:
            norm.update(torch.from_numpy(np.ascontiguousarray(runner.obs)).to(runner.device))
        actions = np.random.randint(0, runner.n_actions, size=A).astype(np.int32)
        actions[t >= steps] = -1
        if hasattr(runner.vec_env, "step_all") and norm is None:
            runner.vec_env.step_all(actions)
        elif hasattr(runner.vec_env, "step_arrays"):
            runner.obs, _, _ = runner.vec_env.step_arrays(actions)
        else:
            runner.obs, _, _, _ = runner.vec_env.step(actions)


class PPO:
    """Thin object form of `train` (the north star's "rl.ppo.PPO"): PPO(model, log).train()."""

    def __init__(self, model, log=None):
        self.model, self.log = model, log or Logger()

    def train(self):
        return train(self.model, self.log)


def train(model, log: Logger):
    start_time = time.time()
    log.add_variable(LogVariable("ep_score", 100, "stats", display_width=12))
    log.add_variable(LogVariable("ep_length", 100, "stats", display_width=12))
    world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
    batch_size = args.n_steps * args.agents * world
    final_epoch = min(args.epochs, args.limit_epochs) if args.limit_epochs is not None else args.epochs
    end_iteration = math.ceil((final_epoch * 1e6) / batch_size)

    runner = Runner(model, log, action_dist="gaussian" if args.env.type == "mujoco" else "discrete")
    runner.vec_env = envs.create_envs_classic(rank=rank, world=world)
    runner.reset()
    log.important("Generated {} agents x {} rank(s) using {} ({:.2f}M params) model.".format(
        args.agents, world, runner.model.name, runner.model.model_size() / 1e6))

    start_iteration = 0
    checkpoints = get_checkpoints(args.log_folder) if args.restore in ("auto", "always") else []
    if args.restore == "always" and not checkpoints:
        raise Exception(f"Error: no restore point at {args.log_folder} found.")
    if checkpoints:
        restored_step = runner.load_checkpoint(os.path.join(args.log_folder, checkpoints[0][1]))
        log.info(" -resumed from step {:.0f}M".format(restored_step / 1e6))
        start_iteration = (restored_step // batch_size) + 1
    else:
        desync_envs(runner, 1, max(args.env.warmup_period, 1))

    next_checkpoint = args.checkpoint_every
    last_print = time.time()
    bench_t0, bench_steps = None, 0
    for iteration in range(start_iteration, end_iteration + 1):
        step_start = time.time()
        env_step = iteration * batch_size
        t0 = time.time()
        runner.generate_rollout()
        torch.cuda.synchronize()
        time_rollout = time.time() - t0
        t0 = time.time()
        runner.calculate_returns()
        torch.cuda.synchronize()
        time_returns = time.time() - t0
        t0 = time.time()
        runner.train()
        torch.cuda.synchronize()
        time_train = time.time() - t0
        stats = runner.fetch_stats()
        step_time = time.time() - step_start
        if iteration > start_iteration and bench_t0 is None:
            bench_t0 = time.time()  # skip the first (warm-up) iteration like the reference's benchmark mode
        elif bench_t0 is not None:
            bench_steps += batch_size
        if not args.disable_logging:
            log.watch("iteration", iteration, display_width=5)
            log.watch("env_step", env_step + batch_size, display_width=12, display_name="step")
            log.watch("walltime", time.time() - start_time, display_width=10)
            log.watch_mean("fps", int(batch_size / max(step_time, 1e-9)))
            log.watch_mean("time_rollout", time_rollout, display_name="t_roll")
            log.watch_mean("time_returns", time_returns, display_name="t_ret")
            log.watch_mean("time_train", time_train, display_name="t_train")
            log.record_step()
            if rank == 0 and (time.time() - last_print > 10 or iteration == end_iteration):
                log.print_variables(include_header=True)
                last_print = time.time()
        if rank == 0 and (env_step + batch_size) >= next_checkpoint:
            runner.save_checkpoint(os.path.join(args.log_folder, "checkpoint-{:03d}M-params.pt".format(
                int((env_step + batch_size) // 1e6))), env_step + batch_size)
            next_checkpoint += args.checkpoint_every
    if args.benchmark_mode and bench_t0 is not None and rank == 0:
        # the reference's benchmark line (rl/ppo.py:354-365)
        print(f"IPS: {bench_steps / max(time.time() - bench_t0, 1e-9):.0f}")
    if rank == 0 and not args.disable_logging:
        log.export_to_csv(os.path.join(args.log_folder, "training_log.csv"))
    return runner


def get_checkpoints(path):
    """Newest first: [(epoch_M, filename)] for files named checkpoint-XXXM-params.pt[.gz] (rl/rollout.py:460-470)."""
    out = []
    if path and os.path.isdir(path):
        for f in os.listdir(path):
            if f.startswith("checkpoint-") and (f.endswith("M-params.pt") or f.endswith("M-params.pt.gz")):
                try:
                    out.append((int(f[len("checkpoint-"):f.index("M-")]), f))
                except ValueError:
                    pass
    return sorted(out, reverse=True)
