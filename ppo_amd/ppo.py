"""Training loop — host mirror of the reference's rl/ppo.py (`train(model, log)` :47-383,
`desync_envs` :21-44).  One iteration = generate_rollout -> calculate_returns -> train, each timed
like the reference (`time_rollout`, `time_returns`, `time_train`, :256-284); throughput is the
reference's IPS = env steps / wall seconds (:354-365), summed over ranks.
"""
import json
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
    (rl/ppo.py:21-44); envs that are done warming up receive action -1 (skip).

    Data-parallel runs always take `max_duration` iterations: the per-rank draw of `steps` differs, and every
    iteration may carry collectives (observation-normaliser moments, the reward normaliser's moments_sync inside
    `vec_env.step`), so a rank-dependent loop length would pair one rank's warm-up reduce with another rank's
    gradient reduce."""
    if max_duration <= 0:
        return
    A = runner.A
    steps = np.random.randint(min_duration, max_duration + 1, size=A)
    norm = getattr(runner.model, "obs_norm", None)
    n_iter = int(max_duration) if runner.world > 1 else int(steps.max())
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
    if hasattr(runner.vec_env, "parts"):
        runner.obs = np.concatenate([p.obs for p in runner.vec_env.parts])


class PPO:
    """Thin object form of `train` (the north star's "rl.ppo.PPO"): PPO(model, log).train()."""

    def __init__(self, model, log=None):
        self.model, self.log = model, log or Logger()

    def train(self):
        return train(self.model, self.log)


def checkpoint_iterations(end_iteration: int, batch_size: int):
    """Iterations after which a checkpoint is written (rl/ppo.py:165-172): every `checkpoint_every` env steps, an
    optional early one at 1M steps, and the last."""
    if args.checkpoint_every == 0:
        return []
    its = [x // batch_size for x in range(0, end_iteration * batch_size + 1, int(args.checkpoint_every))]
    if args.save_early_checkpoint:
        its.append(int(1e6) // batch_size)
    its.append(end_iteration)
    return sorted(set(its))


def checkpoint_name(env_step: int) -> str:
    """rl/utils.py:802-804, 316-317: checkpoint-XXXM-params.pt with the step rounded to millions."""
    return os.path.join(args.log_folder, "checkpoint-{:03.0f}M-params.pt".format(round(env_step / 1e6)))


def train(model, log: Logger):
    start_time = time.time()
    if args.log_folder is None:  # train.py resolves "<output>/<experiment>/<run> [guid]"; direct callers get the plain folder
        args.log_folder = args.output_folder
    log.add_variable(LogVariable("ep_score", 100, "stats", display_width=12))
    log.add_variable(LogVariable("ep_length", 100, "stats", display_width=12))
    world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
    batch_size = args.n_steps * args.agents * world  # env steps per iteration over all ranks (rl/ppo.py:75)
    final_epoch = min(args.epochs, args.limit_epochs) if args.limit_epochs is not None else args.epochs
    end_iteration = math.ceil((final_epoch * 1e6) / batch_size)

    runner = Runner(model, log, action_dist="gaussian" if args.env.type == "mujoco" else "discrete")
    runner.vec_env = envs.create_envs_classic(rank=rank, world=world)
    runner.reset()
    log.important("Generated {} agents x {} rank(s) using {} ({:.2f}M params) model.".format(
        args.agents, world, runner.model.name, runner.model.model_size() / 1e6))

    # ---- detect a previous experiment (rl/ppo.py:92-123)
    checkpoints = runner.get_checkpoints(args.log_folder)
    if args.restore == "always" and not checkpoints:
        raise Exception(f"Error: no restore point at {args.log_folder} found.")
    start_iteration, did_restore = 0, False
    if args.initial_model is not None:
        # load the model but start from step 0 (:101-107)
        runner.load_checkpoint(os.path.join(args.log_folder, args.initial_model))
        runner.step = 0
        log.info(f"Initialized with reference policy {args.initial_model}.")
    elif checkpoints and args.restore in ("auto", "always"):
        log.info("Previous checkpoint detected.")
        restored_step = runner.load_checkpoint(os.path.join(args.log_folder, checkpoints[0][1]))
        log.info(" -resumed from step {:.0f}M".format(restored_step / 1e6))
        start_iteration = (restored_step // batch_size) + 1  # (:114) the reference's own off-by-one, kept
        did_restore = True
    if world > 1:
        runner.sync_replicas()
    if not did_restore:
        desync_envs(runner, 1, max(args.env.warmup_period, 1))
    else:
        desync_envs(runner, 1, 4, verbose=False)  # a few new frames through the wrappers (:134)

    if rank == 0:
        os.makedirs(args.log_folder, exist_ok=True)
        with open(os.path.join(args.log_folder, "params.txt"), "wt") as f:  # (:137-139)
            json.dump({k: v for k, v in args.flatten().items() if isinstance(v, (int, float, str, bool, type(None)))},
                      f, indent=4)
    checkpoint_its = checkpoint_iterations(end_iteration, batch_size)

    iteration = start_iteration
    env_step = start_iteration * batch_size
    if args.save_initial_checkpoint and args.save_checkpoints:
        runner.save_checkpoint(checkpoint_name(env_step), env_step)
    last_print = time.time()
    start_train_time = time.time()
    for _ in range(start_iteration, end_iteration):
        runner.step = iteration * batch_size  # (:252)
        step_start = time.time()
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
        runner.fetch_stats()
        step_time = time.time() - step_start
        iteration += 1
        env_step += batch_size
        if not args.disable_logging:
            log.watch("iteration", iteration, display_width=5)
            log.watch("env_step", env_step, display_width=12, display_name="step")
            log.watch("walltime", time.time() - start_time, display_width=10)
            log.watch_mean("fps", int(batch_size / max(step_time, 1e-9)))
            log.watch_mean("time_rollout", time_rollout, display_name="t_roll")
            log.watch_mean("time_returns", time_returns, display_name="t_ret")
            log.watch_mean("time_train", time_train, display_name="t_train")
            log.record_step()
            if rank == 0 and (time.time() - last_print > args.debug_print_freq or iteration == end_iteration):
                log.print_variables(include_header=True)
                last_print = time.time()
        # periodic checkpoints (:334-339); every rank takes part (the env state is gathered over ranks)
        if args.save_checkpoints and iteration in checkpoint_its and iteration != end_iteration - 1 \
                and (not did_restore or iteration != start_iteration):
            runner.save_checkpoint(checkpoint_name(env_step), env_step)
    if args.benchmark_mode and rank == 0:
        # the reference's benchmark lines (rl/ppo.py:354-365): every iteration of this call, warm-up included
        took = time.time() - start_train_time
        steps = (end_iteration - start_iteration) * batch_size
        print(f"Completed {steps:,} steps in {took:.1f}s")
        print(f"IPS: {round(steps / max(took, 1e-9)):,}")
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
