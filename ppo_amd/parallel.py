"""Data-parallel plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI
on the GPU box, "gloo" in CPU tests).  The reference has no counterpart (no
torch.distributed anywhere); correctness is defined as "an N-rank run equals a 1-rank run with N*A envs
up to minibatch composition and fp32 reduction order" (SURVEY.md §8e).

The PPO path shards by env column, so the exchanges are:
  * the flat fp32 gradient once per optimiser step (4.37 MB for IMPALA), as TWO buckets issued on a
    communication stream as soon as each is final (`GradReducer`): the dense layer + heads (91 % of the
    bytes, final right after the first two launches of the backward pass) travel under the convolution
    backward; the convolution gradients (0.39 MB, final only when the deferred slab reduction has run)
    follow at the end and are latency-bound on xGMI.  The division by world size is folded into the Adam
    kernel (`grad_div`), the global-norm clip happens after the reduction so every rank clips alike;
  * `allreduce_sum_(moments)` once per batch — the three float64 advantage moments {sum, sumsq, n};
  * once at start-up, `broadcast_` of every replica's parameters (and optimiser / normaliser state) from
    rank 0, so replicas are identical whatever their process-local RNG drew.
"""
import hashlib

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def backend_name() -> str:
    return dist.get_backend() if world_size() > 1 else "none"


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def broadcast_(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    if world_size() > 1:
        dist.broadcast(t, src=src)
    return t


def assert_identical_across_ranks(tensors, what: str) -> str:
    """sha256 over the tensors' bytes, compared across ranks; raises on any mismatch.  Returns the digest."""
    h = hashlib.sha256()
    for t in tensors:
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    digest = h.hexdigest()
    if world_size() > 1:
        got = [None] * world_size()
        dist.all_gather_object(got, digest)
        if len(set(got)) != 1:
            raise RuntimeError(f"{what} differ across data-parallel ranks: {got}")
    return digest


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


class GradReducer:
    """Bucketed, overlapped all-reduce of one net's flat gradient buffer.

    `early(stream)` is called by the backward pass (DualHeadNet.grad_ready_hook) on the stream that has just
    queued the last kernel writing the EARLY bucket [split, end) — the dense layer and the heads, which the
    backward pass produces first.  The bucket's all-reduce is issued on the communication stream behind an
    event, so it runs under the convolution backward.  `finish()` (called by Runner.optimizer_step when the
    whole backward is queued) issues the LATE bucket [0, split) and makes the current stream wait for both.
    Sum only: Adam divides by the world size (`grad_div`).

    Without the hook having fired (MLP nets, phases that bypass it) `finish()` reduces the whole buffer in one
    collective, which is the round-1 behaviour.  Every rank issues the same collectives in the same order.

    Exposed communication (the part of the reductions that did not hide under compute) is measured with two
    events per optimiser step on the compute stream: `exposed_ms()` averages them since the last call."""

    def __init__(self, grad: torch.Tensor, split: int):
        self.grad, self.split = grad, int(split)
        self.comm = torch.cuda.Stream(device=grad.device) if grad.is_cuda else None
        self._early_event = torch.cuda.Event() if grad.is_cuda else None
        self._early_work = None
        self._marks = []

    def early(self, stream=None):
        if world_size() == 1 or self.comm is None or not 0 < self.split < self.grad.numel():
            return
        self._early_event.record(stream or torch.cuda.current_stream())
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(self._early_event)
            self._early_work = dist.all_reduce(self.grad[self.split:], op=dist.ReduceOp.SUM, async_op=True)

    def finish(self, measure: bool = True):
        if world_size() == 1:
            return
        main = torch.cuda.current_stream() if self.comm is not None else None
        marks = None
        if measure and main is not None and len(self._marks) < 4096:
            marks = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            marks[0].record(main)
        if self._early_work is None:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)
        else:
            dist.all_reduce(self.grad[:self.split], op=dist.ReduceOp.SUM)  # ordered after the backward on `main`
            self._early_work.wait()  # `main` now also waits for the early bucket
            self._early_work = None
        if marks is not None:
            marks[1].record(main)
            self._marks.append(marks)

    def abandon(self):
        """A minibatch that ends WITHOUT an optimiser step (a hook stopped the epoch): the early bucket's all-reduce
        may still be in flight on the communication stream, and the next backward pass would overwrite grad[split:]
        under it.  Make the current stream wait for it and forget it; the gradient itself is discarded."""
        if self._early_work is not None:
            self._early_work.wait()
            self._early_work = None

    def exposed_ms(self):
        """Mean time the compute stream spent between 'backward queued' and 'gradients reduced', per optimiser
        step since the last call (synchronises on the last event).  None when nothing was measured."""
        if not self._marks:
            return None
        self._marks[-1][1].synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._marks) / len(self._marks)
        self._marks = []
        return ms
