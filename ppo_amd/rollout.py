"""Rollout runner — host mirror of the reference's rl/rollout.py `Runner`: `generate_rollout` (:702-969),
`calculate_returns` (:1182-1285), `train` (:2220-2255) with its policy / value / distil phases
(`train_policy` :1853-1953, `train_value` :1956-2004, `train_distil` :2140-2164), the minibatch functions
(:1331-1449, :1513-1567, :1610-1771), `train_batch` (:2257-2407) and `optimizer_step` (:1287-1321), for
`--model_architecture=single` (PPO) and `dual` (DNA / TVF), discrete and gaussian action distributions.

What changed relative to the reference, and why (MI355X-first):
  * every rollout buffer lives in HBM (`all_obs` [N+1, A, ...], `value`, `log_policy`, `actions`, rewards,
    terminals, TVF values ...); the reference keeps them in host NumPy arrays and re-uploads minibatches
    (rl/rollout.py:189-250, 2349-2372);
  * one device->host copy per env step (the sampled actions) instead of five (:641, 809-813); action
    sampling (Gumbel-max / gaussian) runs on the GPU next to the policy head; with array-stepping envs the
    step is software-pipelined over two env groups;
  * GAE + lambda-returns are one fused HIP scan, TVF returns one HIP kernel; advantage normalisation, the
    minibatch gather, every loss, backward and Adam are HIP kernels (ppo_amd/csrc); minibatches are read
    through an index vector instead of being gathered (only the observations are gathered);
  * statistics are reduced on the device and fetched once per iteration;
  * data parallelism: env columns are sharded over ranks (one process per GPU); the only exchanges are one
    RCCL all-reduce of the flat gradient per optimiser step and one of the three advantage moments per
    batch, so an N-GPU run equals a 1-GPU run with N*A envs up to minibatch composition.
"""
import copy
import math
import os

import numpy as np
import torch

from . import _lib, parallel, value_quality
from .config import args
from .models import AdamState
from .returns import calculate_bootstrapped_returns


# 1: replay the rollout forward of each env group as a hipGraph.  Measured on MI355X / ROCm 7.2: 0.794 ms per
# env step with graphs vs 0.731 ms eager (graph launches of ~25 kernel nodes cost more than the ctypes calls
# they replace and overlap less across the two group streams), so the default is eager.
ROLLOUT_GRAPH = int(os.environ.get("PPO_AMD_ROLLOUT_GRAPH", "0"))
# the synthetic env uploads a group's observations in this many pieces, each as soon as it has been generated (1: one
# copy after the whole group has been stepped).  Measured: 0.491 / 0.474 / 0.487 / 0.524 ms per env step for 1 / 2 / 4 / 8
UPLOAD_CHUNKS = int(os.environ.get("PPO_AMD_UPLOAD_CHUNKS", "2"))
# discrete policies: sample the actions inside the dense + heads launch of the rollout forward (one launch less per group
# and env step; same bits)
FUSE_ACT = int(os.environ.get("PPO_AMD_FUSE_ACT", "1"))


def _p(t):
    return None if t is None else t.data_ptr()


class Optimizer:
    """What the reference's `self.policy_optimizer` etc. stand for here: a net's flat parameter buffer, one
    set of Adam moments over it and the flag group with its hyper-parameters (rl/rollout.py:126-141)."""

    def __init__(self, net, cfg, state=None):
        self.net, self.cfg, self.state = net, cfg, state

    def zero_grad(self, set_to_none=True):
        self.net.grad.zero_()

    def state_dict(self):
        """`torch.optim.Adam.state_dict()` layout, as the reference checkpoints it (rl/rollout.py:412-421)."""
        if self.state is None:
            return self.net.optimizer_state_dict(self.cfg)
        return self.net.adam_state_dict(self.state.exp_avg, self.state.exp_avg_sq, self.state.step, self.cfg)

    def load_state_dict(self, sd):
        if self.state is None:
            self.net.load_optimizer_state_dict(sd)
        else:
            self.state.exp_avg, self.state.exp_avg_sq, self.state.step = self.net.read_adam_state_dict(sd)


class Runner:
    def __init__(self, model, log, name="agent", action_dist="discrete"):
        if action_dist not in ("discrete", "gaussian"):
            raise ValueError(f"Invalid distribution {action_dist}")
        _lib.require_gpu()
        self.lib = _lib.load()
        self.name = name
        self.model = model
        self.policy_net, self.value_net = model.policy_net, model.value_net
        self.net = self.policy_net
        self.dual = model.architecture == "dual"
        self.log = log
        self.action_dist = action_dist
        self.device = self.net.device
        self.step = 0
        self.batch_counter = 0
        self.vec_env = None
        self.force_generic_rollout = False  # tests: the one-group gym-API loop, to compare the pipelined rollout against
        self.N, self.A = args.n_steps, args.agents
        self.state_shape = tuple(model.input_dims)
        self.n_actions = model.actions
        self.VH = self.net.vh
        if self.VH != 1:
            raise NotImplementedError("one extrinsic value head on this path (intrinsic rewards are out of scope)")
        self.world, self.rank = parallel.world_size(), parallel.rank()
        N, A, nA, VH, dev = self.N, self.A, self.n_actions, self.VH, self.device
        gaussian = action_dist == "gaussian"
        obs_dtype = torch.uint8 if model.policy_net.encoder_kind == "impala" and args.env.type != "mujoco" else torch.float32
        # ---- rollout buffers, all resident in HBM (time-major, env index contiguous)
        self.all_obs = torch.zeros((N + 1, A, *self.state_shape), dtype=obs_dtype, device=dev)
        self.value = torch.zeros((N + 1, A, VH), dtype=torch.float32, device=dev)
        self.returns = torch.zeros((N, A, VH), dtype=torch.float32, device=dev)
        if gaussian:
            self.actions = torch.zeros((N, A, nA), dtype=torch.float32, device=dev)
            self.log_pac = torch.zeros((N, A, nA), dtype=torch.float32, device=dev)
        else:
            self.actions = torch.zeros((N, A), dtype=torch.int32, device=dev)
            self.log_pac = torch.zeros((N, A), dtype=torch.float32, device=dev)
        self.ext_rewards = torch.zeros((N, A), dtype=torch.float32, device=dev)
        self.log_policy = torch.zeros((N, A, nA), dtype=torch.float32, device=dev)
        self.raw_policy = torch.zeros((N, A, nA), dtype=torch.float32, device=dev)
        self.terminals = torch.zeros((N, A), dtype=torch.bool, device=dev)
        self.advantage = torch.zeros((N, A), dtype=torch.float32, device=dev)
        self.raw_advantage = self.advantage
        self.norm_advantage = torch.zeros((N, A), dtype=torch.float32, device=dev)
        # ---- host staging (pinned): one step of actions down, a whole rollout of rewards/dones up
        self._actions_host = torch.zeros((A, nA) if gaussian else (A,),
                                         dtype=torch.float32 if gaussian else torch.int32).pin_memory()
        self._rewards_host = torch.zeros((N, A), dtype=torch.float32).pin_memory()
        self._dones_host = torch.zeros((N, A), dtype=torch.uint8).pin_memory()
        self.obs = None  # current observation (host), set by reset()
        self.all_time = np.zeros((N + 1, A), np.int32)  # env time of every recorded state (rl/rollout.py:808 all_time)
        self._finished_lengths = [[] for _ in range(N)]  # lengths of the episodes that ended at each env step
        self.time = np.zeros(A, np.int32)
        self.episode_score = np.zeros(A, np.float32)
        self.episode_len = np.zeros(A, np.int32)
        self.ep_count = 0
        # ---- TVF (rl/rollout.py:311-313)
        self.tvf = None
        if model.tvf_fixed_head_horizons is not None:
            from .tvf import TVFRunnerModule
            self.tvf = TVFRunnerModule(self)
            self._tvf_weights_dev = torch.as_tensor(np.asarray(self.tvf_weights, np.float32), device=dev)
            # value-phase weights: duplicate-head weights times the optional h_weighting (rl/tvf.py:51-62)
            self._tvf_value_weights_dev = torch.as_tensor(self.tvf.value_loss_weights(), device=dev)
            self._ext_estimate = torch.zeros((N + 1, A), dtype=torch.float32, device=dev)
        # ---- optimisers (rl/rollout.py:126-141)
        self.policy_optimizer = Optimizer(self.policy_net, args.policy_opt)
        self.value_optimizer = Optimizer(self.value_net, args.value_opt) if self.dual else self.policy_optimizer
        # as the reference (rl/rollout.py:145-148): present whenever the distil phase has its own optimiser configured,
        # whatever the architecture (only `dual` ever steps it; its moment buffers are allocated on first use)
        own_distil = args.distil_opt.epochs > 0 and not args.distil.use_policy_opt
        self.distil_optimizer = Optimizer(self.policy_net, args.distil_opt, AdamState()) if own_distil else None
        # ---- device scratch
        self._moments = torch.zeros(3, dtype=torch.float64, device=dev)
        self._moments_ws = torch.zeros(self.lib.ppo_moments_workspace_bytes() // 8, dtype=torch.float64, device=dev)
        self._mean_std = torch.zeros(2, dtype=torch.float32, device=dev)
        self._grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._sample_calls = 0
        # base seed of the counter-based device generators (action sampling, horizon dropout): --seed when given,
        # otherwise one draw from the run's host RNG (the reference's unseeded default gives a different stream per
        # run, rl/config.py:749); saved in checkpoints so that a resumed run continues the same streams
        self._device_seed = int(args.seed) if args.seed >= 0 else int(np.random.randint(1, 2**31 - 1))
        self._graphs = {}
        self._step_events = []
        self._phase_stats = {}
        self.timers = {}
        self._reducers = {}
        if self.world > 1:
            # one base seed for all ranks (the rank is mixed in where draws are made): an unseeded run would otherwise
            # checkpoint rank 0's draw and hand it to every rank on resume, i.e. switch the other ranks' streams
            seed_t = torch.tensor([self._device_seed], dtype=torch.int64, device=dev)
            parallel.broadcast_(seed_t)
            self._device_seed = int(seed_t.item())
            self.sync_replicas()
            for net in {id(n): n for n in (self.policy_net, self.value_net)}.values():
                red = parallel.GradReducer(net.grad, net.early_grad_offset)
                self._reducers[id(net)] = red
                net.grad_ready_hook = red.early

    def sync_replicas(self):
        """Data-parallel ranks must hold the same model: the only per-step exchange is the gradient all-reduce, so
        replicas that start apart stay apart, silently.  Parameters, Adam moments and the observation normaliser
        are broadcast from rank 0 (each rank initialised from its own process RNG when --seed is unset, and a
        restored checkpoint is read by every rank on its own), then a digest is compared across ranks."""
        nets = list({id(n): n for n in (self.policy_net, self.value_net)}.values())
        tensors = []
        for net in nets:
            tensors.append(net.flat)
            for t in (net.exp_avg, net.exp_avg_sq):
                if t is not None:
                    tensors.append(t)
        if self.distil_optimizer is not None and self.distil_optimizer.state.exp_avg is not None:
            tensors += [self.distil_optimizer.state.exp_avg, self.distil_optimizer.state.exp_avg_sq]
        norm = self.model.obs_norm
        if norm is not None:
            tensors += [norm.mean, norm.var, norm.mu, norm.std]
        for t in tensors:
            parallel.broadcast_(t)
        for net in nets:
            net.mark_weights_changed()
        counters = torch.tensor([float(n._adam_step) for n in nets] + [float(norm.count) if norm is not None else 0.0],
                                dtype=torch.float64, device=self.device)
        parallel.broadcast_(counters)
        for n, c in zip(nets, counters[:len(nets)].tolist()):
            n._adam_step = int(c)
        if norm is not None:
            norm.count = float(counters[-1])
        return parallel.assert_identical_across_ranks(tensors, "model replicas")

    # ------------------------------------------------------------------ helpers
    def _call(self, fn_name, *a):
        rc = getattr(self.lib, fn_name)(*a, _lib.current_stream())
        if rc != 0:
            _lib.check(rc, fn_name)

    @property
    def ext_value(self):
        return self.value[:, :, 0]

    @property
    def ext_returns(self):
        return self.returns[:, :, 0]

    @property
    def prev_obs(self):
        return self.all_obs[:-1]

    def anneal(self, x, mode: str = "linear"):
        """rl/rollout.py:331-355: scale x by the remaining (linear) or elapsed (linear_inc / quad_inc) fraction
        of training, measured in env steps over `anneal_target_epoch or epochs` million."""
        assert mode in ["off", "linear", "cos", "cos_linear", "linear_inc", "quad_inc"], f"invalid mode {mode}"
        frac = self.step / ((args.anneal_target_epoch or args.epochs) * 1e6)
        factor = 1.0
        if mode in ("linear", "cos_linear"):
            factor *= float(np.clip(1 - frac, 0, 1))
        if mode == "linear_inc":
            factor *= float(np.clip(frac, 0, 1))
        if mode == "quad_inc":
            factor *= float(np.clip(frac ** 2, 0, 1))
        if mode in ("cos", "cos_linear"):
            factor *= (1 + math.cos(math.pi * 2 * self.step / 20e6)) / 2  # the reference's fixed 20M-step period (:345)
        return x * factor

    def _lr(self, cfg):
        """Learning rate of one optimiser group, annealed linearly to 0 when its lr_anneal flag is set
        (rl/rollout.py:357-392)."""
        return self.anneal(cfg.lr, mode="linear" if cfg.lr_anneal else "off")

    @property
    def policy_lr(self):
        return self._lr(args.policy_opt)

    @property
    def value_lr(self):
        return self._lr(args.value_opt)

    @property
    def distil_lr(self):
        return self._lr(args.distil_opt)

    @property
    def ppo_epsilon(self):
        return self.anneal(args.ppo_epsilon, mode="linear" if args.ppo_epsilon_anneal else "off")

    @property
    def current_entropy_bonus(self):
        """rl/rollout.py:1568-1587: optional rescaling by the size of the action set, optional annealing."""
        if args.entropy_scaling == "off":
            bonus = args.entropy_bonus
        elif args.entropy_scaling == "average":
            assert args.entropy_scaling_base_actions > 0
            bonus = args.entropy_bonus * (args.entropy_scaling_base_actions / self.model.actions)
        elif args.entropy_scaling == "uniform":
            assert args.entropy_scaling_base_actions > 0
            bonus = args.entropy_bonus * (math.log(args.entropy_scaling_base_actions) / math.log(self.model.actions))
        else:
            raise ValueError(f"Invalid entropy_scaling method {args.entropy_scaling}.")
        return self.anneal(bonus) if args.entropy_anneal else bonus

    @property
    def current_advantage_epsilon(self):
        """rl/rollout.py:1589-1594."""
        if args.advantage_epsilon_anneal_factor > 0:
            return args.advantage_epsilon * max((1 / args.advantage_epsilon_anneal_factor) ** (self.step / 10e6), 1e-8)
        return args.advantage_epsilon

    @property
    def value_heads(self):
        return ["ext"]

    @property
    def tvf_horizons(self):
        return self.model.tvf_fixed_head_horizons

    @property
    def tvf_weights(self):
        """Loss weight per TVF head: duplicates removed when spacing the heads (rl/rollout.py:1323-1328)."""
        return np.asarray(self.model.tvf_fixed_head_weights, dtype=np.float32).copy()

    @property
    def K(self):
        return len(self.tvf_horizons)

    def get_current_actions_std(self):
        return 0.0 if self.action_dist == "discrete" else torch.exp(self.policy_net.params["log_std"])

    def reset(self):
        """rl/rollout.py:520-556: reset envs and per-env bookkeeping."""
        assert self.vec_env is not None, "Please assign vec_env first."
        self.obs = self.vec_env.reset()
        if self.tvf is not None:
            self.tvf.episode_length_buffer.clear()
            self.tvf.episode_length_buffer.append(1000)  # rl/rollout.py:553-555
        self.time[:] = 0
        self.episode_score[:] = 0
        self.episode_len[:] = 0
        self.step = 0

    # ------------------------------------------------------------------ rollout
    def _forward_heads(self, obs, tag):
        pol, val = self.policy_net, self.value_net
        hp = pol.heads(pol.encode(obs, train=False, tag=tag), tag)
        hv = val.heads(val.encode(obs, train=False, tag=tag), tag) if self.dual else hp
        return hp, hv

    def _rollout_graph(self, i, B, stream):
        """Opt-in (PPO_AMD_ROLLOUT_GRAPH=1): hipGraph of one env group's policy forward (encoder + heads,
        ~25 launches) from a fixed staging buffer to the fixed head-row buffer.  Weights are read in place, so
        the graph stays valid across optimiser steps.  Returns None (eager path) when off or unavailable."""
        key = (i, B)
        if key not in self._graphs:
            graph = None
            if ROLLOUT_GRAPH:
                try:
                    stage = torch.zeros((B, *self.state_shape), dtype=self.all_obs.dtype, device=self.device)
                    tag = f"i{i}"
                    self._forward_heads(stage, tag)  # warm-up: scratch buffers, kernel attributes
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=stream):
                        hp, hv = self._forward_heads(stage, tag)
                    graph = (g, stage, hp, hv)
                except Exception as e:  # capture not supported: keep the eager path
                    self.log.warn(f"rollout hipGraph capture failed ({type(e).__name__}: {e}); running eagerly")
            self._graphs[key] = graph
        return self._graphs[key]

    def _policy_step(self, t, lo=0, hi=None, tag="i", graph=None):
        """Forward + action sampling for envs [lo, hi) at env step t; writes those columns of row t of the
        rollout buffers.  The sampling counter is keyed by (rollout, t, env, action), so splitting the envs
        into groups does not change which action any env takes.  Dual architecture: the policy comes from
        policy_net, value (and TVF) estimates from value_net (rl/models.py:809-821)."""
        pol, val = self.policy_net, self.value_net
        hi = self.A if hi is None else hi
        B = hi - lo
        A, nA = self.A, self.n_actions
        final = t >= self.N
        seed = self._device_seed * 1000003 + self.rank
        counter = ((self._sample_calls + t) * A + lo) * nA
        first = t * A + lo  # first (t, env) element of this group's rows: pointer arithmetic, no tensor slicing

        def row(buf):
            if final:
                return None
            return buf.data_ptr() + first * buf.stride(1) * buf.element_size()

        values = None if self.dual else self.value.data_ptr() + first * self.value.stride(1) * 4
        discrete = self.action_dist == "discrete"
        if graph is not None:
            g, _stage, hp, hv = graph
            g.replay()
        else:
            if discrete and FUSE_ACT:
                # the sampling rides on the policy net's dense + heads launch when that forward replays its recorded
                # launch list (models.DualHeadNet.encode); otherwise the tail comes back and is launched below
                pol.act_tail = (nA, 1.0, seed & (2**64 - 1), counter, row(self.log_policy), row(self.actions),
                                row(self.log_pac), row(self.raw_policy), values, self.VH)
            hp, hv = self._forward_heads(self.all_obs[t, lo:hi], tag)
        sampled = discrete and graph is None and FUSE_ACT and pol.act_tail is None
        pol.act_tail = None

        if sampled:
            pass
        elif discrete:
            self._call("ppo_policy_act_f32", _p(hp), B, pol.nh, nA, 1.0, None, seed & (2**64 - 1), counter, 0,
                       row(self.log_policy), row(self.actions), row(self.log_pac), row(self.raw_policy), values, self.VH)
        else:
            self._call("ppo_gaussian_act_f32", _p(hp), B, pol.nh, nA, _p(pol.params["log_std"]), None,
                       seed & (2**64 - 1), counter, 0, row(self.actions), row(self.log_pac), row(self.raw_policy),
                       values, self.VH)
        if self.dual:
            self.value[t, lo:hi].copy_(hv[:, val.col_value:val.col_value + self.VH])
        if self.tvf is not None:
            rec = self.tvf.tvf_untrimmed_value  # == tvf_value unless --tvf_trimming
            rec[t, lo:hi].copy_(hv[:, val.col_tvf:].view(B, val.K, self.VH))
            rec[t, lo:hi, 0].zero_()  # the first value head (h = 0) is zero by definition (rl/rollout.py:791)

    def _log_finished(self, finished, ep_len, ep_score, t=None):
        if finished.any():
            self.ep_count += int(finished.sum())
            if t is not None:
                self._finished_lengths[t].extend(int(x) for x in np.asarray(ep_len)[finished])
            for s, l in zip(ep_score[finished][:8], ep_len[finished][:8]):
                if getattr(self, "_pending_episodes", None) is not None:
                    self._pending_episodes.append((s, l))  # logged once the rollout stands (generate_rollout)
                    continue
                self.log.watch_full("ep_score", s, history_length=100)
                self.log.watch_full("ep_length", l, history_length=100)

    def generate_rollout(self):
        """Fill the rollout buffers with N steps from A envs (rl/rollout.py:702-969).

        With a `SplitVecEnv` of array-stepping parts the step is software-pipelined: the GPU runs the policy
        for one group of envs while the host steps the other group, so neither waits for the whole of the
        other (the reference overlaps nothing: rl/rollout.py:792-816 is forward -> sync -> step)."""
        assert self.vec_env is not None, "Please attach vector environment first."
        N, A = self.N, self.A
        self._finished_lengths = [[] for _ in range(N)]
        env = self.vec_env
        parts = [env] if self.force_generic_rollout else getattr(env, "parts", [env])
        if not self.force_generic_rollout and all(hasattr(p, "step_arrays") for p in parts):
            # array-stepping groups: the synthetic env, and gym-API envs behind the process pool and its vector
            # wrappers (ppo_amd/hybrid_vec_env.py `PoolGroup`, ppo_amd/wrappers.py `parts`)
            nets = list({id(n): n for n in (self.policy_net, self.value_net)}.values())
            armed = [n for n in nets if n.chain_split_armed()]
            # the two-workgroups-per-image launch (models.CHAIN_SPLIT) can, in principle, time out waiting for a partner
            # workgroup; its results are then void.  An env that can be put back exactly is, and the rollout is redone
            snap = self._rollout_snapshot(env) if armed and getattr(env, "exact_snapshot", False) else None
            self._pending_episodes = []
            self._rollout_pipelined(parts, armed)
            if armed and any(n.chain_split_error() for n in armed):
                self._chain_split_failed(armed, env, snap, parts)
            for s_, l_ in self._pending_episodes:
                self.log.watch_full("ep_score", s_, history_length=100)
                self.log.watch_full("ep_length", l_, history_length=100)
            self._pending_episodes = None
            if hasattr(env, "finish_rollout"):
                # what the vector wrappers do to the rewards needs every env's reward of a step at once (the running
                # return statistics): applied here, step by step in the reference's order, once the rollout is in
                env.finish_rollout(self._rewards_host.numpy(), self._dones_host.numpy())
        else:
            self._rollout_generic(env)
        self.ext_rewards.copy_(self._rewards_host, non_blocking=True)
        self.terminals.view(torch.uint8).copy_(self._dones_host, non_blocking=True)
        if self.tvf is not None:
            self.tvf.apply_trimming(self.all_time, self._finished_lengths)  # no-op unless --tvf_trimming
        self._sample_calls += N + 1
        self.step += N * A * self.world

    def _rollout_snapshot(self, env):
        """Everything a pipelined rollout changes on the host, for an env whose save_state is complete."""
        from . import checkpoint
        from .wrappers import VecWrapper
        layers, inner = [], env
        while isinstance(inner, VecWrapper):
            layers.append((inner, inner.snapshot_state()))
            inner = inner.__dict__.get("env")
        snap = {"env": copy.deepcopy(checkpoint.save_env_state(inner)), "inner": inner, "layers": layers,
                "time": self.time.copy(),
                "episode_score": self.episode_score.copy(), "episode_len": self.episode_len.copy(),
                "ep_count": self.ep_count, "np_random": np.random.get_state()}
        if self.tvf is not None:
            snap["episode_length_buffer"] = list(self.tvf.episode_length_buffer)
        return snap

    def _chain_split_failed(self, nets, env, snap, parts):
        """A workgroup of the two-workgroups-per-image launch gave up waiting for its partner's half of a map: the head
        outputs of this rollout (and the actions drawn from them) cannot be trusted.  The one-workgroup launch takes
        over for the rest of the run, and the rollout is generated again: from the saved start state where the env
        can be put back exactly (same bytes as a run with PPO_AMD_CHAIN_SPLIT=0: the sampling counter has not
        moved), otherwise from where the envs are now (their N steps are lost, nothing else)."""
        from . import checkpoint
        for n in nets:
            n.chain_split_disable()
        self.log.warn("the split chained launch timed out waiting for a partner workgroup: switched to one workgroup "
                      "per image for the rest of the run and redoing this rollout"
                      + (" from its start state" if snap is not None else " from the envs' current state"))
        if snap is not None:
            for wrapper, state in snap["layers"]:
                wrapper.restore_snapshot(state)
            checkpoint.restore_env_state(snap["inner"], snap["env"])
            self.time = snap["time"].copy()
            self.episode_score[:] = snap["episode_score"]
            self.episode_len[:] = snap["episode_len"]
            self.ep_count = snap["ep_count"]
            np.random.set_state(snap["np_random"])
            if self.tvf is not None:
                self.tvf.episode_length_buffer.clear()
                self.tvf.episode_length_buffer.extend(snap["episode_length_buffer"])
            self._pending_episodes = []
        elif hasattr(env, "finish_rollout"):
            # the lost steps were real env steps: the vector wrappers' statistics take them in as usual
            env.finish_rollout(self._rewards_host.numpy().copy(), self._dones_host.numpy().copy())
        self._finished_lengths = [[] for _ in range(self.N)]
        self._rollout_pipelined(parts, [])

    def _rollout_pipelined(self, parts, split_nets=()):
        """`split_nets`: the networks whose forwards may take the two-workgroups-per-image launch during this rollout
        (the caller checks their error word afterwards)."""
        self._split_nets_active = list(split_nets)
        try:
            self._rollout_pipelined_impl(parts)
        finally:
            for n in split_nets:
                n.allow_chain_split = False

    def _rollout_pipelined_impl(self, parts):
        N = self.N
        self.all_time[0] = self.time
        rew_np, done_np = self._rewards_host.numpy(), self._dones_host.numpy()
        act_np = self._actions_host.numpy()
        bounds = np.cumsum([0] + [p.num_envs for p in parts]).tolist()
        assert bounds[-1] == self.A
        P = len(parts)
        if len(self._step_events) < P:
            self._step_events = [torch.cuda.Event() for _ in range(P)]
            self._copy_events = [torch.cuda.Event() for _ in range(P)]
            self._copy_stream = torch.cuda.Stream(device=self.device)
            self._part_streams = [torch.cuda.Stream(device=self.device) for _ in range(P)]
        events, copy_events, copy_stream = self._step_events, self._copy_events, self._copy_stream
        main = torch.cuda.current_stream()
        # each env group has its own compute stream (and scratch-buffer set), so the policy steps of the two
        # groups overlap on the GPU: at half the batch most kernels expose fewer workgroups than there are CUs
        streams = self._part_streams if P > 1 else [main]
        copy_stream.wait_stream(main)  # earlier readers of all_obs (the previous train phase) are done
        for s_ in streams:
            if s_ is not main:
                s_.wait_stream(main)
        graphs = [self._rollout_graph(i, bounds[i + 1] - bounds[i], streams[i]) if P > 1 else None for i in range(P)]
        for n in self._split_nets_active:  # (after the graph captures: a recorded graph keeps the one-workgroup launch)
            n.allow_chain_split = True
        norm = self.model.obs_norm

        # a group may be stepped (and uploaded) in several pieces: a leaf's H2D runs while the next leaf is stepped
        leaves = []
        for i in range(P):
            lo, row = bounds[i], []
            for leaf in getattr(parts[i], "leaves", [parts[i]]):
                row.append((leaf, lo, lo + leaf.num_envs))
                lo += leaf.num_envs
            leaves.append(row)

        def upload(i, t, only=None):
            # H2D of group i's observations (pinned -> HBM) into row t on the copy stream, so it runs on a DMA engine
            # under the other group's policy step.  (set_stream rather than the `with torch.cuda.stream` context: this
            # runs 2 x 257 times per rollout and the host is on the critical path; the caller restores the stream)
            torch.cuda.set_stream(copy_stream)
            for leaf, lo, hi in (leaves[i] if only is None else [only]):
                if graphs[i] is not None:  # the graph reads a fixed staging buffer; the rollout row is a D2D copy of it
                    graphs[i][1][lo - bounds[i]:hi - bounds[i]].copy_(leaf.obs_t, non_blocking=True)
                else:
                    self.all_obs[t, lo:hi].copy_(leaf.obs_t, non_blocking=True)

        def enqueue(i, t):
            # policy + sampling for group i at step t behind its upload (issued when its envs were stepped), actions D2H;
            # all async
            lo, hi = bounds[i], bounds[i + 1]
            if graphs[i] is not None:
                torch.cuda.set_stream(copy_stream)
                copy_events[i].record()
                self.all_obs[t, lo:hi].copy_(graphs[i][1], non_blocking=True)
            torch.cuda.set_stream(streams[i])
            streams[i].wait_event(copy_events[i])
            self._policy_step(t, lo, hi, tag=tags[i], graph=graphs[i])
            if t < N:
                host_rows[i].copy_(self.actions[t, lo:hi], non_blocking=True)
                events[i].record()

        def step_envs(i, t):
            # the one host wait of group i's step: its actions have landed; then step it on host cores, leaf by leaf,
            # each leaf's next observations going up as soon as they exist
            events[i].synchronize()
            for leaf, lo, hi in leaves[i]:
                if norm is None and graphs[i] is None and UPLOAD_CHUNKS > 1 and hasattr(leaf, "step_upload"):
                    # stepping and upload in one call: each quarter of the group goes up while the rest is stepped
                    torch.cuda.set_stream(copy_stream)
                    leaf.step_upload(act_np[lo:hi], rew_np[t, lo:hi], done_np[t, lo:hi], self.all_obs[t + 1, lo:hi],
                                     copy_stream, UPLOAD_CHUNKS)
                    continue
                leaf.step_arrays(act_np[lo:hi], rew_np[t, lo:hi], done_np[t, lo:hi])
                if norm is None:
                    upload(i, t + 1, only=(leaf, lo, hi))
            if norm is None:
                copy_events[i].record()  # (current stream = the copy stream: upload() left it so)
            lo, hi = bounds[i], bounds[i + 1]
            time_now, ep_len, ep_score = parts[i].last_episode_stats
            done = done_np[t, lo:hi].astype(bool)
            self.all_time[t + 1, lo:hi] = parts[i].landed_time(done)
            self._log_finished(done, ep_len, ep_score, t)

        tags = [f"i{i}" if P > 1 else "i" for i in range(P)]
        host_rows = [self._actions_host[bounds[i]:bounds[i + 1]] for i in range(P)]
        if norm is not None:
            # Observation normalisation: the running statistics take in EVERY env's observation of step t before any
            # group's forward of step t (rl/rollout.py:735-741), so the groups cannot run a step apart.  Per step: both
            # uploads, one statistics update on the first group's stream, then the groups' policy steps on their own
            # streams; the host steps group i while the GPU still runs the later groups' policy steps.  (The generic
            # path synchronised the whole device once per env step and overlapped nothing.)
            if not hasattr(self, "_norm_event"):
                self._norm_event = torch.cuda.Event()
            try:
                for t in range(N + 1):
                    for i in range(P):
                        upload(i, t)
                    copy_events[0].record()
                    torch.cuda.set_stream(streams[0])
                    streams[0].wait_event(copy_events[0])
                    if t < N:
                        norm.update(self.all_obs[t])
                    self._norm_event.record()
                    for i in range(P):
                        torch.cuda.set_stream(streams[i])
                        streams[i].wait_event(self._norm_event)
                        self._policy_step(t, bounds[i], bounds[i + 1], tag=tags[i])
                        if t < N:
                            host_rows[i].copy_(self.actions[t, bounds[i]:bounds[i + 1]], non_blocking=True)
                            events[i].record()
                    if t < N:
                        for i in range(P):
                            step_envs(i, t)
                        # the next step's upload overwrites nothing the GPU still reads, but its statistics update
                        # must follow this step's forwards (they read mu / std): order stream 0 behind the others
                        for s_ in streams[1:]:
                            streams[0].wait_stream(s_)
            finally:
                torch.cuda.set_stream(main)
            for s_ in streams:
                if s_ is not main:
                    main.wait_stream(s_)
            main.wait_stream(copy_stream)
            self.time = self.all_time[N].copy()
            self.obs = parts[0].obs if P == 1 else np.concatenate([p.obs for p in parts])
            return
        try:
            for i in range(P):  # the observations the envs hold now are row 0; later rows go up from step_envs
                upload(i, 0)
                copy_events[i].record()
            for t in range(N + 1):
                for i in range(P):
                    enqueue(i, t)  # needs obs(i, t): group i was stepped for t-1 below
                    j, tj = (i + 1) % P, (t if i + 1 == P else t - 1)
                    if 0 <= tj < N:
                        step_envs(j, tj)  # overlaps the GPU work just queued for group i
        finally:
            torch.cuda.set_stream(main)
        for s_ in streams:
            if s_ is not main:
                main.wait_stream(s_)
        main.wait_stream(copy_stream)  # the graph path's last all_obs rows are written by D2D copies queued there
        self.time = self.all_time[N].copy()
        self.obs = parts[0].obs if P == 1 else np.concatenate([p.obs for p in parts])

    def _rollout_generic(self, env):
        """gym-API vector env (`step(actions) -> obs, rew, done, infos`), one group.  Also the path taken with
        observation normalisation: the running statistics are updated from ALL envs' observations of step t
        before that step's forward, as the reference does (rl/rollout.py:735-741), which the group-pipelined
        rollout cannot express."""
        N = self.N
        norm = self.model.obs_norm
        rew_np, done_np = self._rewards_host.numpy(), self._dones_host.numpy()
        act_np = self._actions_host.numpy()
        stream = torch.cuda.current_stream()
        for t in range(N + 1):
            self.all_time[t] = self.time
            # zero-copy view: with the process pool self.obs IS the shared pinned block, so this is one async H2D
            src = self.obs if self.obs.flags.c_contiguous else np.ascontiguousarray(self.obs)
            self.all_obs[t].copy_(torch.from_numpy(src), non_blocking=True)
            if norm is not None and t < N:
                norm.update(self.all_obs[t])
            self._policy_step(t)
            if t == N:
                break  # final state: only its value estimate is needed (rl/rollout.py:871-878)
            self._actions_host.copy_(self.actions[t], non_blocking=True)
            stream.synchronize()  # the one device->host sync of the step: the envs need the actions
            self.obs, rew, dones, infos = env.step(act_np.copy())
            rew_np[t] = rew
            done_np[t] = dones
            self.time = np.asarray([i.get("time", 0) for i in infos], np.int32)  # rl/rollout.py:753
            self._log_finished(np.asarray(dones, bool), np.asarray([i.get("ep_length", 0) for i in infos]),
                               np.asarray([i.get("ep_score", 0.0) for i in infos]), t)

    @torch.no_grad()
    def detached_batch_forward(self, obs, aux_features=None, max_batch_size=None, **kwargs):
        """Forward a large batch in chunks, results concatenated (rl/rollout.py:557-598)."""
        max_batch_size = max_batch_size or args.max_micro_batch_size
        obs = self.model.prep_for_model(obs)
        chunks = []
        for i in range(0, obs.shape[0], max_batch_size):
            out = self.model.forward(obs[i:i + max_batch_size], **kwargs)
            chunks.append({k: v.clone() for k, v in out.items()})
        return {k: torch.cat([c[k] for c in chunks], dim=0) for k in chunks[0]}

    # ------------------------------------------------------------------ returns
    def calculate_returns(self):
        """Advantages (lambda_policy) and value targets (lambda_value) in one fused scan
        (rl/rollout.py:1182-1285 -> rl/returns.py:7-67), then the TVF return targets (rl/tvf.py:210-271)."""
        N, A = self.N, self.A
        if self.tvf is not None:
            self._ext_estimate.copy_(self.tvf.get_tvf_ext_value_estimate(new_gamma=args.gamma))
            value = self._ext_estimate
        else:
            value = self.value.view(N + 1, A)
        self._call("ppo_gae_scan_f32", _p(self.ext_rewards), _p(value), _p(value[N]), _p(self.terminals),
                   _lib.PPO_TERM_U8, _p(self.advantage), _p(self.returns), N, A, A, float(args.gamma),
                   float(args.lambda_policy), float(args.lambda_value), _lib.PPO_SCAN_AUTO)
        if self.tvf is not None:
            self.tvf.tvf_returns[..., 0].copy_(self.tvf.calculate_tvf_returns(value_head="ext"))
        if not args.disable_logging:
            self.log_returns(value)

    @property
    def reward_scale(self):
        """What the env stack multiplied the rewards by (rl/rollout.py:1793-1802)."""
        if args.env.reward_normalization == "off":
            return 1.0
        if args.env.reward_normalization != "rms":
            raise ValueError(f"Invalid reward normalization {args.env.reward_normalization}")
        from . import wrappers
        norm = wrappers.get_wrapper(self.vec_env, wrappers.VecNormalizeRewardWrapper)
        return 1.0 if norm is None else float(1.0 / norm.std)

    def log_returns(self, value_estimate):
        """The diagnostics the reference writes at the end of calculate_returns (rl/rollout.py:1199, 1252-1285):
        moments of the value estimates / advantages / returns every batch; feature statistics and explained
        variance every 4th batch unless --disable_ev."""
        N = self.N
        head0 = self.value_heads[0]
        named = [("*ext_value_estimates", value_estimate, {}), ("adv_ext", self.advantage, {"display_width": 0})]
        for i, head in enumerate(self.value_heads):
            named.append((f"*return_{head}", self.returns[..., i], {"display_width": 0}))
            named.append((f"value_{head}", self.value[..., i], {"display_name": "v_" + head}))
        value_quality.log_batch_moments(self.log, named)
        self.log.watch_mean("reward_scale", self.reward_scale, display_width=0, history_length=1)
        self.log.watch_mean("entropy_bonus", self.current_entropy_bonus, display_width=0, history_length=1)
        self.log.watch("*gamma", args.gamma)
        if self.tvf is not None:
            self.log.watch("*tvf_gamma", args.tvf.gamma)
            self.log.watch_stats("*tvf_return_ext", self.tvf.tvf_returns[:, :, -1].cpu().numpy())
        if args.disable_ev or self.batch_counter % 4 != 3:
            return
        value_quality.log_feature_statistics(
            self.log, self.detached_batch_forward(self.all_obs[0], output="full", include_features=True))
        ext_value = self.value[..., self.value_heads.index("ext")] if "ext" in self.value_heads else self.value[..., 0]
        targets = calculate_bootstrapped_returns(self.ext_rewards, self.terminals, ext_value[N], float(args.gamma))
        if self.tvf is not None:
            self.tvf.log_tvf_curve_quality(ext_value[:N], targets)
        else:
            value_quality.log_dna_value_quality(self.log, ext_value[:N], targets)

    # ------------------------------------------------------------------ training
    def _normalize_advantages(self):
        """(a - mean) / (std + eps) over the whole (global) batch (rl/rollout.py:1887-1900)."""
        n = self.N * self.A
        self._call("ppo_moments_f64", _p(self.advantage), n, _p(self._moments), _p(self._moments_ws))
        parallel.allreduce_sum_(self._moments)
        self._call("ppo_normalize_f32", _p(self.advantage), n, _p(self._moments), float(self.current_advantage_epsilon),
                   _p(self.norm_advantage), _p(self._mean_std))
        if args.advantage_clipping is not None:
            self.norm_advantage.clamp_(-args.advantage_clipping, args.advantage_clipping)

    def optimizer_step(self, optimizer=None, label="policy", norm_out=None):
        """All-reduce (DP) + global-norm clip + Adam in the flat buffer (rl/rollout.py:1287-1321)."""
        opt = optimizer or self.policy_optimizer
        net, cfg = opt.net, opt.cfg
        red = self._reducers.get(id(net))
        if red is not None:
            red.finish()  # late bucket + join of the early one that left during the backward pass
        net.adam_step(lr=self._lr(cfg), beta1=cfg.adam_beta1, beta2=cfg.adam_beta2, eps=cfg.adam_epsilon,
                      max_grad_norm=args.max_grad_norm if args.grad_clip_mode == "global_norm" else 0.0,
                      grad_div=float(self.world), grad_norm_out=self._grad_norm if norm_out is None else norm_out,
                      state=opt.state)
        return self._grad_norm if norm_out is None else norm_out

    def _micro_batches(self, mb, force_micro_batch_size=None):
        """(micro-batch size, count) of a per-rank minibatch of mb samples (rl/rollout.py:2310-2316): the reference
        splits a minibatch into passes of at most --max_micro_batch_size samples whose gradients accumulate."""
        micro = force_micro_batch_size if force_micro_batch_size is not None else min(args.max_micro_batch_size, mb)
        if micro <= 0 or mb % micro:
            raise ValueError(f"minibatch {mb} is not a multiple of the micro-batch size {micro}")
        return micro, mb // micro

    class _Accumulator:
        """Gradient accumulation over the micro-batches of one minibatch.  Every backward pass of the HIP path
        OVERWRITES net.grad, so passes 2.. are added into a second flat buffer (ppo_accumulate_f32) and the sum is
        copied back before the optimiser step; with one micro-batch nothing happens.  The data-parallel early-bucket
        hook is parked meanwhile: only the accumulated gradient may leave."""

        def __init__(self, runner, net, count):
            self.r, self.net, self.count, self.k = runner, net, count, 0
            self.hook = net.grad_ready_hook
            if count > 1:
                net.grad_ready_hook = None
                self.acc = net._buf("grad_accum", tuple(net.grad.shape))

        def after_backward(self):
            if self.count > 1:
                if self.k == 0:
                    self.acc.copy_(self.net.grad)
                else:
                    self.r._call("ppo_accumulate_f32", _p(self.acc), _p(self.net.grad), self.net.grad.numel())
            self.k += 1

        def finish(self):
            if self.count > 1:
                self.net.grad.copy_(self.acc)
                self.net._presummed = 0  # the last pass's sums of g^2 are not the accumulated gradient's
            self.restore_hook()

        def restore_hook(self):
            """Also the exit path of a failed step: the parked hook must come back whatever happened."""
            self.net.grad_ready_hook = self.hook

    def _run_epochs(self, label, optimizer, epochs, mini_batch_size, step_fn, n_stats):
        """Permutation minibatching over the rollout (rl/rollout.py:2257-2407): per epoch one host shuffle
        (np.random, as the reference :2319-2320); per minibatch — in micro-batches of at most --max_micro_batch_size
        samples, gradients accumulated, each pass scaled by 1 / micro_batches (:2331-2374) — gather the observations,
        run step_fn(mb_obs, index, loss_scale) -> per-sample stats [n, n_stats], then step the optimiser; the stats
        are column-summed into one device row per minibatch."""
        B = self.N * self.A
        net = optimizer.net
        mb = parallel.local_minibatch(mini_batch_size)  # the flag is the GLOBAL minibatch
        if B % mb:
            raise ValueError(f"batch {B} is not a multiple of the per-rank minibatch {mb}")
        n_mb = B // mb
        micro, n_micro = self._micro_batches(mb)
        obs_rows = self.all_obs[:self.N].view(B, -1)
        row_bytes = obs_rows.shape[1] * obs_rows.element_size()
        # MLP nets on the fused path read their rows of the whole batch through the permutation and column-sum the
        # statistics in their weight-gradient launch: no gather launch, no column-sum launch
        fused = bool(getattr(net, "mlp_fused", False)) and net.obs_norm is None and self.all_obs.dtype == torch.float32
        # ... and the IMPALA nets read uint8 image rows through the permutation in their first convolution
        in_conv = not fused and hasattr(net, "takes_obs_index") and net.takes_obs_index(self.all_obs)
        obs_all = self.all_obs[:self.N].view(B, *self.state_shape)
        mb_obs = None if (fused or in_conv) else net._buf("mb_obs", (micro, *self.state_shape), self.all_obs.dtype)
        stat_rows = net._buf(f"stat_rows_{label}", (epochs * n_mb, n_stats))
        norm_rows = net._buf(f"norm_rows_{label}", (epochs * n_mb,))
        k = 0
        for _epoch in range(epochs):
            ordering = np.arange(B, dtype=np.int32)
            np.random.shuffle(ordering)
            order_dev = torch.from_numpy(ordering).to(self.device, non_blocking=True)
            for j in range(n_mb):
                acc = self._Accumulator(self, net, n_micro)
                try:
                    for u in range(n_micro):
                        idx = order_dev[j * mb + u * micro:j * mb + (u + 1) * micro]
                        if fused:
                            step_fn(obs_rows, idx, 1.0 / n_micro, stat_sums=stat_rows[k], stat_accumulate=bool(u))
                            acc.after_backward()
                            continue
                        if in_conv:
                            stats = step_fn(obs_all, idx, 1.0 / n_micro)
                        else:
                            self._call("ppo_gather_rows", _p(obs_rows), row_bytes, B, _p(idx), micro, _p(mb_obs))
                            stats = step_fn(mb_obs, idx, 1.0 / n_micro)
                        acc.after_backward()
                        self._call("ppo_colsum_f32", _p(stats), micro, n_stats, n_stats, _p(stat_rows[k]), 1 if u else 0)
                    acc.finish()
                finally:
                    acc.restore_hook()
                self.optimizer_step(optimizer, label, norm_out=norm_rows[k:k + 1])  # the norm lands in its row: no copy launch
                k += 1
        self._phase_stats[label] = (stat_rows, norm_rows, mb)

    def train_policy(self):
        """PPO epochs over the rollout (rl/rollout.py:1853-1953); the single architecture trains the value
        head in the same pass (:1744-1746)."""
        if args.policy_opt.epochs == 0:
            return
        B = self.N * self.A
        net = self.policy_net
        self._normalize_advantages()
        returns = None if self.dual else self.returns
        net.zero_untouched_grads()
        if self.action_dist == "discrete":
            def step(mb_obs, idx, loss_scale, **kw):
                return net.ppo_minibatch(mb_obs, self.actions, self.log_pac, self.log_policy, self.norm_advantage,
                                         returns, eps_clip=self.ppo_epsilon, ent_coef=self.current_entropy_bonus,
                                         vf_coef=args.ppo_vf_coef, loss_scale=loss_scale, index=idx, **kw)
        else:
            actions, log_pac = self.actions.view(B, self.n_actions), self.log_pac.view(B, self.n_actions)

            def step(mb_obs, idx, loss_scale, **kw):
                return net.gaussian_minibatch(mb_obs, actions, log_pac, self.norm_advantage, returns,
                                              eps_clip=self.ppo_epsilon, vf_coef=args.ppo_vf_coef, loss_scale=loss_scale,
                                              index=idx, **kw)
        self._run_epochs("policy", self.policy_optimizer, args.policy_opt.epochs, args.policy_opt.mini_batch_size,
                         step, 8)

    def train_value(self):
        """Value phase of the dual architecture (rl/rollout.py:1956-2004): value_net regresses the value
        targets, and the TVF heads their truncated-return targets."""
        if args.value_opt.epochs == 0:
            return
        B = self.N * self.A
        net = self.value_net
        use_ext = self.tvf is None or args.tvf.include_ext
        returns = self.returns.view(B, self.VH) if use_ext else None
        tvf_returns = self.tvf.tvf_returns[:, :, :, -1].reshape(B, self.K) if self.tvf is not None else None
        weights = self._tvf_value_weights_dev if self.tvf is not None else None
        net.zero_untouched_grads()

        keep = 1.0 - args.tvf.horizon_dropout if self.tvf is not None else 1.0
        seed = self._device_seed * 1000003 + 7919 * (self.rank + 1)

        def step(mb_obs, idx, loss_scale, **kw):
            off = self.tvf.next_dropout_offset(idx.shape[0] * self.K) if keep < 1.0 else 0
            return net.value_minibatch(mb_obs, returns=returns, tvf_returns=tvf_returns, tvf_weights=weights,
                                       vf_coef=args.ppo_vf_coef, tvf_coef=args.tvf.coef, loss_scale=loss_scale, index=idx,
                                       tvf_keep_prob=keep, dropout_seed=seed, dropout_offset=off, **kw)
        self._run_epochs("value", self.value_optimizer, args.value_opt.epochs, args.value_opt.mini_batch_size, step, 4)

    def wants_distil_update(self, location=None):
        """rl/rollout.py:2211-2218."""
        return (self.dual and args.distil_opt.epochs > 0 and self.step >= args.distil.delay * 1e6
                and self.batch_counter % args.distil.period == args.distil.period - 1
                and (location is None or location == args.distil.order))

    def get_distil_batch(self, samples_wanted=None):
        """Targets and the policy to stay close to, from the rollout (rl/rollout.py:2050-2112, the
        replay-free path): value_net's estimates recorded during the rollout, and either the rollout's
        policy (`before_policy`) or the just-updated policy re-evaluated over the batch (`after_policy`)."""
        N, A = self.N, self.A
        B = N * A
        if samples_wanted not in (None, B) or (0 < args.distil.batch_size != B):
            raise NotImplementedError("distillation from a replay buffer is out of scope; distil_batch_size = rollout")
        if args.distil.target != "value" or args.distil.loss != "kl_policy" or args.distil.value_loss != "mse":
            raise NotImplementedError("distil target 'value', loss 'kl_policy', value_loss 'mse' are built")
        use_tvf = self.tvf is not None and not args.distil.force_ext
        if use_tvf and 0 < args.distil.max_heads < self.K:
            raise NotImplementedError("distil_max_heads sub-sampling is not built (default: all heads)")
        batch = {"use_tvf": use_tvf}
        if use_tvf:
            batch["distil_targets"] = self.tvf.tvf_untrimmed_value[:N, :, :, 0].reshape(B, self.K)
        else:
            batch["distil_targets"] = self.value[:N].reshape(B)
        key = "raw_policy" if self.action_dist == "gaussian" else "log_policy"
        if args.distil.order == "before_policy":
            old = getattr(self, key).view(B, self.n_actions)
        else:
            net = self.policy_net
            old = net._buf("distil_old_policy", (B, self.n_actions))
            obs = self.all_obs[:N].view(B, *self.state_shape)
            chunk = max(args.max_micro_batch_size, 1)
            for i in range(0, B, chunk):
                out = net.forward(obs[i:i + chunk], exclude_value=True)
                old[i:i + chunk].copy_(out[key])
        batch["old_policy"] = old
        return batch

    def train_distil(self):
        """Distillation phase (rl/rollout.py:2140-2164): policy_net's value (or TVF) heads learn value_net's
        estimates under a KL constraint that keeps the policy where it is."""
        if args.distil_opt.epochs == 0:
            return
        batch = self.get_distil_batch()
        net = self.policy_net
        gaussian = self.action_dist == "gaussian"
        weights = self._tvf_weights_dev if batch["use_tvf"] else None
        net.zero_untouched_grads()

        def step(mb_obs, idx, loss_scale, **kw):
            return net.distil_minibatch(mb_obs, batch["distil_targets"], batch["old_policy"], beta=args.distil.beta,
                                        use_tvf=batch["use_tvf"], weights=weights, gaussian=gaussian,
                                        loss_scale=loss_scale, index=idx, **kw)
        opt = self.policy_optimizer if args.distil.use_policy_opt else self.distil_optimizer
        self._run_epochs("distil", Optimizer(net, args.distil_opt, opt.state), args.distil_opt.epochs,
                         args.distil_opt.mini_batch_size, step, 4)

    def train(self):
        """rl/rollout.py:2220-2255: [distil] -> policy -> (dual) value -> [distil]."""
        self._phase_stats = {}
        if self.wants_distil_update("before_policy"):
            self.train_distil()
        self.train_policy()
        if self.dual:
            self.train_value()
            if self.wants_distil_update("after_policy"):
                self.train_distil()
        self.batch_counter += 1

    # ---- the reference's dict-based minibatch API (rl/rollout.py:1331, 1513, 1610, 2257), for callers that
    # drive the phases themselves; `data` holds minibatch-sized device tensors
    def train_policy_minibatch(self, data, loss_scale=1.0):
        net = self.policy_net
        returns = data.get("returns") if not self.dual else None
        if self.action_dist == "discrete":
            stats = net.ppo_minibatch(data["prev_state"], data["actions"].to(torch.int32), data["log_pac"],
                                      data.get("log_policy"), data["advantages"], returns, eps_clip=self.ppo_epsilon,
                                      ent_coef=self.current_entropy_bonus, vf_coef=args.ppo_vf_coef,
                                      loss_scale=loss_scale)
        else:
            stats = net.gaussian_minibatch(data["prev_state"], data["actions"], data["log_pac"], data["advantages"],
                                           returns, eps_clip=self.ppo_epsilon, vf_coef=args.ppo_vf_coef,
                                           loss_scale=loss_scale)
        s = stats.double().mean(0).cpu().numpy()
        return {"loss": float(-s[6] * loss_scale), "kl_approx": float(s[4]), "kl_true": float(s[5]), "clip_frac": float(s[3])}

    def train_value_minibatch(self, data, loss_scale=1.0, single_value_head=None):
        if single_value_head is not None:
            raise NotImplementedError("training a single TVF head (noise-scale estimation) is out of scope")
        weights = self._tvf_value_weights_dev if "tvf_returns" in data else None
        stats = self.value_net.value_minibatch(data["prev_state"], returns=data.get("returns"),
                                               tvf_returns=data.get("tvf_returns"), tvf_weights=weights,
                                               vf_coef=args.ppo_vf_coef, tvf_coef=args.tvf.coef, loss_scale=loss_scale)
        total = stats[:, 2].double() * loss_scale
        return {"loss": float(total.mean()), "loss_std": float(total.std())}

    def train_distil_minibatch(self, data, loss_scale=1.0, **kwargs):
        use_tvf = self.tvf is not None and not args.distil.force_ext
        gaussian = self.action_dist == "gaussian"
        stats = self.policy_net.distil_minibatch(
            data["prev_state"], data["distil_targets"], data["old_raw_policy" if gaussian else "old_log_policy"],
            beta=args.distil.beta, use_tvf=use_tvf, weights=self._tvf_weights_dev if use_tvf else None,
            gaussian=gaussian, loss_scale=loss_scale)
        total = stats[:, 2].double() * loss_scale
        return {"loss": float(total.mean()), "loss_std": float(total.std())}

    def train_batch(self, batch_data, mini_batch_func, mini_batch_size, optimizer, label, epoch=None, hooks=None,
                    thinning=1.0, force_micro_batch_size=None, delta_threshold=None):
        """One epoch of permutation minibatches through `mini_batch_func(data, loss_scale=...)`, the reference's
        dict-based API (rl/rollout.py:2257-2407): minibatches are split into micro-batches of `force_micro_batch_size`
        (default min(--max_micro_batch_size, minibatch)) samples whose gradients accumulate under
        loss_scale = 1 / micro_batches; `thinning` keeps that fraction of every micro-batch; `hooks`
        ("after_micro_batch"(context), "after_mini_batch"(context) -> truthy stops before the optimiser step) are
        called as the reference calls them; entries whose name starts with '*' are passed through whole.  Returns
        {'mini_batches', 'outputs'[, 'did_break']}."""
        if delta_threshold is not None and delta_threshold > 0:
            raise Exception("Not supported")
        assert "prev_state" in batch_data, "Batches must contain 'prev_state' field of dims (B, *state_shape)"
        B = len(batch_data["prev_state"])
        for k_, v in batch_data.items():
            if not k_.startswith("*"):
                assert len(v) == B, f"Batch input must all match in entry count. Expecting {B} but found {len(v)} on {k_}"
        mb = parallel.local_minibatch(mini_batch_size)
        assert B % mb == 0
        n_mb = B // mb
        micro, n_micro = self._micro_batches(mb, force_micro_batch_size)
        ordering = np.arange(B)
        np.random.shuffle(ordering)
        data = {k_: (v.to(self.device) if isinstance(v, torch.Tensor) else torch.as_tensor(v).to(self.device))
                for k_, v in batch_data.items()}
        net = optimizer.net
        outputs, context, counter = [], {}, 0
        for j in range(n_mb):
            optimizer.zero_grad()
            acc = self._Accumulator(self, net, n_micro)
            try:
                for u in range(n_micro):
                    sample = ordering[counter * micro:(counter + 1) * micro]
                    counter += 1
                    if thinning < 1.0:
                        sample = sample[:int(micro * thinning)]
                    idx = torch.from_numpy(sample).to(self.device)
                    micro_context = {"epoch": epoch, "mini_batch": j, "micro_batch": u, "is_first": j == 0,
                                     "is_last": j == n_mb - 1}
                    micro_data = {"context": micro_context}
                    for k_, v in data.items():
                        micro_data[k_] = v if k_.startswith("*") else v[idx].contiguous()
                    outputs.append(mini_batch_func(micro_data, loss_scale=1 / n_micro))
                    acc.after_backward()
                    if hooks is not None and "after_micro_batch" in hooks:
                        hooks["after_micro_batch"](micro_context)
                acc.finish()
            finally:
                acc.restore_hook()
            context = {"mini_batches": j + 1, "outputs": outputs}
            if hooks is not None and "after_mini_batch" in hooks:
                stop = bool(hooks["after_mini_batch"](context))
                if self.world > 1:
                    # a stop decided from rank-local data (a KL threshold, say) must be everyone's: otherwise one rank
                    # leaves while the others issue the late-bucket all-reduce and the collective sequences diverge
                    flag = torch.tensor([1.0 if stop else 0.0], device=self.device)
                    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
                    stop = bool(flag.item() > 0)
                if stop:
                    red = self._reducers.get(id(net))
                    if red is not None:
                        red.abandon()  # the early bucket already left during the backward pass: join it before leaving
                    context["did_break"] = True
                    break
            self.optimizer_step(optimizer, label)
        return context

    def fetch_stats(self):
        """ONE device->host copy per phase and iteration with everything the reference logs per minibatch
        (rl/rollout.py:1685-1691, 1759-1769, 1317, 1427-1446, 1563)."""
        out = {}
        if "policy" in self._phase_stats:
            stat_rows, norm_rows, mb = self._phase_stats["policy"]
            s = stat_rows.cpu().numpy().astype(np.float64) / mb
            out.update({"loss_pg": s[:, 0].mean(), "entropy": s[:, 1].mean(), "loss_v_ext": s[:, 2].mean(),
                        "clip_frac": s[:, 3].mean(), "kl_approx": s[:, 4].mean(), "kl_true": s[:, 5].mean(),
                        "loss_policy": s[:, 6].mean(), "grad_policy": float(norm_rows.mean().item()),
                        "adv_mean": float(self._mean_std[0].item()), "adv_std": float(self._mean_std[1].item())})
        if "value" in self._phase_stats:
            stat_rows, norm_rows, mb = self._phase_stats["value"]
            s = stat_rows.cpu().numpy().astype(np.float64) / mb
            out.update({"loss_v_ext": s[:, 0].mean(), "loss_tvf": s[:, 1].mean(), "loss_value": s[:, 2].mean(),
                        "grad_value": float(norm_rows.mean().item())})
        if "distil" in self._phase_stats:
            stat_rows, norm_rows, mb = self._phase_stats["distil"]
            s = stat_rows.cpu().numpy().astype(np.float64) / mb
            out.update({"loss_distil_value": s[:, 0].mean(), "loss_distil_policy": s[:, 1].mean(),
                        "loss_distil": s[:, 2].mean(), "distil_mse": s[:, 3].mean(),
                        "grad_distil": float(norm_rows.mean().item())})
        if not args.disable_logging:
            for k_, v in out.items():
                self.log.watch_mean(k_, v)
        return out

    # ------------------------------------------------------------------ checkpoints (rl/rollout.py:394-517)
    def get_checkpoints(self, path):
        """[(epoch_M, filename)] newest first (rl/rollout.py:460-470)."""
        from .ppo import get_checkpoints
        return get_checkpoints(path)

    def save_checkpoint(self, filename, step, disable_log=False, disable_replay=False, disable_env_state=False,
                        disable_optimizer=False):
        """Model under the reference's state_dict names, optimiser states, counters, env / wrapper state;
        gzip container when --checkpoint_compression (file name + '.gz').  Returns the path written.

        Data parallel: replicas hold the same model, so rank 0 writes it; what differs per rank — its env columns'
        state, its reward normaliser, its host RNG (minibatch permutations) — is gathered to rank 0 and stored as one
        entry per rank.  Every rank must call this (it is a collective when world > 1)."""
        from . import checkpoint
        local = {"np_random": np.random.get_state(), "ep_count": self.ep_count, "time": self.time.copy(),
                 "episode_score": self.episode_score.copy(), "episode_len": self.episode_len.copy()}
        if self.tvf is not None:
            local["episode_length_buffer"] = [int(x) for x in self.tvf.episode_length_buffer]  # rl/rollout.py:399
            local["tvf_dropout_calls"] = int(self.tvf._dropout_calls)  # a resumed run must not replay dropout masks
        if not disable_env_state and self.vec_env is not None:
            local["env_state"] = checkpoint.save_env_state(self.vec_env)
        per_rank = [local]
        if self.world > 1:
            per_rank = [None] * self.world if self.rank == 0 else None
            torch.distributed.gather_object(checkpoint.to_plain(local), per_rank, dst=0)
        if self.rank != 0:
            return None
        # the reference's top-level keys (rl/rollout.py:396-407; tests/golden/checkpoint_golden.json is their tree) ...
        data = {"step": step, "ep_count": self.ep_count, "batch_counter": self.batch_counter,
                "episode_length_buffer": [int(x) for x in self.tvf.episode_length_buffer] if self.tvf is not None else [1000],
                "model_state_dict": dict(self.model.state_dict()), "reward_scale": float(self.reward_scale) if self.vec_env is not None else 1.0,
                "episode_score": self.episode_score.copy(),
                # ... and this package's own: the data-parallel layout and the device generators' position
                "world": self.world, "sample_calls": self._sample_calls, "device_seed": self._device_seed,
                "rank_state": per_rank}
        if not disable_optimizer:  # torch.optim.Adam.state_dict() layout each; the reference writes the value
            data["policy_optimizer_state_dict"] = self.policy_optimizer.state_dict()  # optimiser's even when it is
            data["value_optimizer_state_dict"] = self.value_optimizer.state_dict()    # the policy optimiser (single)
            if self.distil_optimizer is not None:
                data["distil_optimizer_state_dict"] = self.distil_optimizer.state_dict()
        if self.model.obs_norm is not None:
            data["obs_rms"] = self.model.obs_norm.state_dict()  # rl/rollout.py:438-439
        return checkpoint.save(data, filename, bool(args.checkpoint_compression))

    def load_checkpoint(self, checkpoint_path):
        """Restores model, optimisers, counters and this rank's env / RNG state; returns the env step
        (rl/rollout.py:472-517).  Every rank reads the file itself."""
        from . import checkpoint
        cp = checkpoint.load(checkpoint_path)
        self.model.load_state_dict(cp["model_state_dict"])
        for key, opt in (("policy_optimizer_state_dict", self.policy_optimizer),
                         ("value_optimizer_state_dict", self.value_optimizer if self.dual else None),
                         ("distil_optimizer_state_dict", self.distil_optimizer)):
            if opt is not None and key in cp:
                opt.load_state_dict(cp[key])
        if self.model.obs_norm is not None:
            self.model.obs_norm.load_state_dict(cp["obs_rms"])  # rl/rollout.py:511-513
        self.step = cp["step"]
        self.ep_count = cp.get("ep_count", 0)
        self.batch_counter = cp.get("batch_counter", 0)
        self._sample_calls = cp.get("sample_calls", 0)
        self._device_seed = int(cp.get("device_seed", self._device_seed))
        ranks = cp.get("rank_state") or []
        if len(ranks) == self.world:
            mine = ranks[self.rank]
            if mine.get("np_random") is not None:
                np.random.set_state(mine["np_random"])
            self.ep_count = mine.get("ep_count", self.ep_count)
            if mine.get("time") is not None:
                self.time = np.asarray(mine["time"], np.int32).copy()
            for key in ("episode_score", "episode_len"):
                if mine.get(key) is not None:
                    getattr(self, key)[:] = mine[key]
            if self.tvf is not None and mine.get("episode_length_buffer") is not None:
                self.tvf.episode_length_buffer.clear()
                self.tvf.episode_length_buffer.extend(mine["episode_length_buffer"])
            if self.tvf is not None:
                self.tvf._dropout_calls = int(mine.get("tvf_dropout_calls", self.tvf._dropout_calls))
            if mine.get("env_state") and self.vec_env is not None:
                checkpoint.restore_env_state(self.vec_env, mine["env_state"])
                if hasattr(self.vec_env, "parts"):
                    self.obs = np.concatenate([p.obs for p in self.vec_env.parts])
                elif hasattr(self.vec_env, "obs"):
                    self.obs = self.vec_env.obs
        elif ranks:
            self.log.warn(f"checkpoint was written by {len(ranks)} rank(s), this run has {self.world}: model and "
                          "optimiser restored, env / RNG state not")
        return self.step
