"""Data-parallel plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI
on the GPU box, "gloo" in CPU tests).  The reference has no counterpart (no
torch.distributed anywhere); correctness is defined as "an N-rank run equals a 1-rank run with N*A envs
up to minibatch composition and fp32 reduction order" (SURVEY.md §8e).

The PPO path shards by env column, so there are exactly two exchanges:
  * `allreduce_sum_(flat_grad)` once per optimiser step — ONE flat fp32 buffer (4.37 MB for IMPALA),
    a single collective, latency-bound on xGMI; the division by world size is folded into the Adam
    kernel (`grad_div`), the global-norm clip happens after the reduction so every rank clips alike;
  * `allreduce_sum_(moments)` once per batch — the three float64 advantage moments {sum, sumsq, n}.
"""
import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def shard(total_envs_per_rank: int):
    """(first global env index, count) of this rank's env columns: contiguous blocks in rank order."""
    return rank() * total_envs_per_rank, total_envs_per_rank


def local_minibatch(global_minibatch: int) -> int:
    """`--policy_opt_mini_batch_size` is the GLOBAL minibatch; each rank contributes 1/world of it, so the
    number of optimiser steps per batch matches a 1-rank run with world*A envs."""
    w = world_size()
    if global_minibatch % w:
        raise ValueError(f"global minibatch {global_minibatch} is not divisible by world size {w}")
    return global_minibatch // w


def mean_var_from_moments(moments):
    """(mean, population variance) from all-reduced {sum, sumsq, n}."""
    s, ss, n = (float(x) for x in moments)
    mean = s / n
    return mean, max(ss / n - mean * mean, 0.0)
