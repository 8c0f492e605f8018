#!/usr/bin/env python3
"""Entry point, drop-in for the reference's train.py (`python train.py --flag=value ...`,
train.py:86-210): parse rl.config flags, pick the device, seed, build the model, run ppo.train.

One process per GPU.  Single GPU: `python train.py ...`.  N GPUs of one node:
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 train.py ...`
(`--agents` is then the per-rank env count; gradients are all-reduced over RCCL).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ppo_amd import envs, logger, models, ppo  # noqa: E402
from ppo_amd.config import args  # noqa: E402


def make_model(log):
    """train.py:33-82 of the reference: model from the env's spaces and the model flags."""
    from ppo_amd import tvf
    obs_shape, n_actions = envs.get_env_spec()
    horizons = weights = None
    if args.tvf.enabled:
        horizons, weights = tvf.get_value_head_horizons(args.tvf.value_heads, args.tvf.max_horizon,
                                                        args.tvf.head_spacing, include_weight=True)
        args.tvf.value_heads = len(horizons)  # duplicate horizons are merged (train.py:50-52)
    return models.TVFModel(
        encoder=args.model.encoder, encoder_args=args.model.encoder_args, input_dims=obs_shape, actions=n_actions,
        device=args.device, architecture=args.model.architecture, dtype=torch.float32,
        hidden_units=args.model.hidden_units,
        encoder_activation_fn="tanh" if args.env.type == "mujoco" else "relu",
        tvf_fixed_head_horizons=horizons, tvf_fixed_head_weights=weights,
        tvf_feature_sparsity=args.tvf.feature_sparsity, tvf_feature_window=args.tvf.feature_window,
        head_scale=args.model.head_scale,
        head_bias=args.model.head_bias, value_head_names=("ext",),
        observation_normalization=args.observation_normalization,
        freeze_observation_normalization=args.freeze_observation_normalization,
        norm_eps=args.observation_normalization_epsilon,
        # train.py:166-178: low / medium let the GPU use reduced-precision convolution arithmetic (there TF32, here the
        # split-bf16 launches of the 32-channel stacks); high is exact float32 everywhere
        precision=args.precision)


def get_previous_experiment_guid(experiment_path, run_name):
    """Folder "<run_name> [<guid8>]" of an earlier run of this experiment, if any (train.py:11-19)."""
    if not os.path.exists(experiment_path):
        return None
    for f in sorted(os.listdir(experiment_path)):
        if f[:-(8 + 3)] == run_name and f.endswith("]"):
            return f[-(8 + 1):-1]
    return None


def resolve_log_folder(rank=0, world=1):
    """args.log_folder = "<output_folder>/<experiment_name>/<run_name> [<guid8>]" (train.py:142-154, 187): with
    --restore=auto|always an earlier run's folder is reused, so its checkpoints are found.  Rank 0 decides and
    every rank uses its answer."""
    import uuid
    if args.log_folder is None:
        guid, error = None, None
        if rank == 0:
            if args.restore in ("always", "auto"):
                guid = get_previous_experiment_guid(os.path.join(args.output_folder, args.experiment_name), args.run_name)
                if guid is None and args.restore == "always":
                    error = (f"Could not restore experiment {args.experiment_name}:{args.run_name}. "
                             "Previous run not found.")
            guid = guid or uuid.uuid4().hex[-8:]
        if world > 1:  # the error travels with the answer: every rank leaves together, none waits in the broadcast
            box = [guid, error]
            torch.distributed.broadcast_object_list(box, src=0)
            guid, error = box
        if error:
            raise SystemExit(error)
        args.log_folder = "{} [{}]".format(os.path.join(args.output_folder, args.experiment_name, args.run_name), guid)
    if rank == 0:
        os.makedirs(args.log_folder, exist_ok=True)
    if world > 1:
        torch.distributed.barrier()
    return args.log_folder


def main():
    args.setup()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # before any HIP call, env worker or pinned buffer: this rank's share of the cores of its GPU's NUMA node
    # (PPO_AMD_AFFINITY=0 leaves the mask alone)
    from ppo_amd import affinity
    affinity.pin_rank(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if not torch.cuda.is_available():
        raise SystemExit("train.py: no HIP device visible; this build has no CPU path")
    torch.cuda.set_device(local)
    args.device = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
    log = logger.Logger(quiet=(int(os.environ.get("RANK", "0")) != 0))
    if args._ignored:
        log.warn(f"ignoring flags of subsystems outside the PPO hot path: {args._ignored}")
    if args.seed >= 0:  # train.py:157-163
        torch.manual_seed(args.seed)
        np.random.seed(args.seed + int(os.environ.get("RANK", "0")))
    resolve_log_folder(int(os.environ.get("RANK", "0")), world)
    log.info("Logging to folder " + args.log_folder)
    model = make_model(log)
    try:
        return ppo.train(model, log)
    finally:
        if world > 1:
            torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
