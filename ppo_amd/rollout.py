"""Rollout runner — host mirror of the reference's rl/rollout.py `Runner` for the PPO hot path
(`--model_architecture=single`): `generate_rollout` (:702-969), `calculate_returns` (:1182-1285),
`train` / `train_policy` / `train_batch` (:2220-2255, :1853-1953, :2257-2407), `optimizer_step`
(:1287-1321).

What changed relative to the reference, and why (MI355X-first):
  * every rollout buffer lives in HBM (`all_obs` uint8 [N+1, A, C, H, W], `value`, `log_policy`,
    `actions`, rewards, terminals ...); the reference keeps them in host NumPy arrays and re-uploads
    minibatches (rl/rollout.py:189-250, 2349-2372);
  * one device->host copy per env step (the sampled actions) instead of five (:641, 809-813);
    action sampling (Gumbel-max) runs on the GPU next to the policy head;
  * GAE + lambda-returns are one fused HIP scan; advantage normalisation, the minibatch gather,
    the PPO loss, backward and Adam are HIP kernels (ppo_amd/csrc);
  * statistics are reduced on the device and fetched once per iteration;
  * data parallelism: env columns are sharded over ranks (one process per GPU); the only exchanges
    are one RCCL all-reduce of the flat gradient per optimiser step and one of the three advantage
    moments per batch, so an N-GPU run equals a 1-GPU run with N*A envs up to minibatch composition.
"""
import time

import numpy as np
import torch

from . import _lib, parallel
from .config import args


def _p(t):
    return None if t is None else t.data_ptr()


class Runner:
    def __init__(self, model, log, name="agent", action_dist="discrete"):
        if action_dist != "discrete":
            raise NotImplementedError("the HIP path implements the discrete-action PPO update")
        if args.model.architecture != "single":
            raise NotImplementedError("Runner here implements --model_architecture=single (PPO); see DESIGN.md")
        _lib.require_gpu()
        self.lib = _lib.load()
        self.name = name
        self.model = model
        self.net = model.policy_net
        self.log = log
        self.action_dist = action_dist
        self.device = self.net.device
        self.step = 0
        self.batch_counter = 0
        self.vec_env = None
        self.N, self.A = args.n_steps, args.agents
        self.state_shape = tuple(model.input_dims)
        self.n_actions = model.actions
        self.VH = self.net.vh
        self.world, self.rank = parallel.world_size(), parallel.rank()
        N, A, nA, VH, dev = self.N, self.A, self.n_actions, self.VH, self.device
        obs_dtype = torch.float32 if args.env.type == "mujoco" else torch.uint8
        # ---- rollout buffers, all resident in HBM (time-major, env index contiguous)
        self.all_obs = torch.zeros((N + 1, A, *self.state_shape), dtype=obs_dtype, device=dev)
        self.value = torch.zeros((N + 1, A, VH), dtype=torch.float32, device=dev)
        self.returns = torch.zeros((N, A, VH), dtype=torch.float32, device=dev)
        self.actions = torch.zeros((N, A), dtype=torch.int32, device=dev)
        self.ext_rewards = torch.zeros((N, A), dtype=torch.float32, device=dev)
        self.log_policy = torch.zeros((N, A, nA), dtype=torch.float32, device=dev)
        self.raw_policy = torch.zeros((N, A, nA), dtype=torch.float32, device=dev)
        self.log_pac = torch.zeros((N, A), dtype=torch.float32, device=dev)
        self.terminals = torch.zeros((N, A), dtype=torch.bool, device=dev)
        self.advantage = torch.zeros((N, A), dtype=torch.float32, device=dev)
        self.raw_advantage = self.advantage
        self.norm_advantage = torch.zeros((N, A), dtype=torch.float32, device=dev)
        # ---- host staging (pinned): one step of actions down, a whole rollout of rewards/dones up
        self._actions_host = torch.zeros(A, dtype=torch.int32).pin_memory()
        self._rewards_host = torch.zeros((N, A), dtype=torch.float32).pin_memory()
        self._dones_host = torch.zeros((N, A), dtype=torch.uint8).pin_memory()
        self.obs = None  # current observation (host, pinned), set by reset()
        self.time = np.zeros(A, np.int32)
        self.episode_score = np.zeros(A, np.float32)
        self.episode_len = np.zeros(A, np.int32)
        self.ep_count = 0
        # ---- device scratch
        self._moments = torch.zeros(3, dtype=torch.float64, device=dev)
        self._moments_ws = torch.zeros(self.lib.ppo_moments_workspace_bytes() // 8, dtype=torch.float64, device=dev)
        self._mean_std = torch.zeros(2, dtype=torch.float32, device=dev)
        self._grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._sample_calls = 0
        self._step_events = []
        self._stat_rows = None
        self.timers = {}

    # ------------------------------------------------------------------ helpers
    def _call(self, fn_name, *a):
        rc = getattr(self.lib, fn_name)(*a, _lib.current_stream())
        if rc != 0:
            _lib.check(rc, fn_name)

    @property
    def ext_value(self):
        return self.value[:, :, 0]

    @property
    def prev_obs(self):
        return self.all_obs[:-1]

    @property
    def ppo_epsilon(self):
        return args.ppo_epsilon

    @property
    def current_entropy_bonus(self):
        return args.entropy_bonus

    def reset(self):
        """rl/rollout.py:520-556: reset envs and per-env bookkeeping."""
        assert self.vec_env is not None, "Please assign vec_env first."
        self.obs = self.vec_env.reset()
        self.time[:] = 0
        self.episode_score[:] = 0
        self.episode_len[:] = 0
        self.step = 0

    # ------------------------------------------------------------------ rollout
    def _policy_step(self, t, lo=0, hi=None):
        """Forward + action sampling for envs [lo, hi) at env step t; writes those columns of row t of the
        rollout buffers.  The sampling counter is keyed by (rollout, t, env, action), so splitting the envs
        into groups does not change which action any env takes."""
        net = self.net
        hi = self.A if hi is None else hi
        acts = net.encode(self.all_obs[t, lo:hi], train=False)
        heads = net.heads(acts["h"], "i")
        A, nA = self.A, self.n_actions
        final = t >= self.N
        seed = (int(args.seed) if args.seed >= 0 else 0) * 1000003 + self.rank
        counter = ((self._sample_calls + t) * A + lo) * nA
        self._call("ppo_policy_act_f32", _p(heads), hi - lo, net.nh, nA, 1.0, None, seed & (2**64 - 1), counter, 0,
                   None if final else _p(self.log_policy[t, lo:hi]), None if final else _p(self.actions[t, lo:hi]),
                   None if final else _p(self.log_pac[t, lo:hi]), None if final else _p(self.raw_policy[t, lo:hi]),
                   _p(self.value[t, lo:hi]), self.VH)

    def _log_finished(self, finished, ep_len, ep_score):
        if finished.any():
            self.ep_count += int(finished.sum())
            for s, l in zip(ep_score[finished][:8], ep_len[finished][:8]):
                self.log.watch_full("ep_score", s, history_length=100)
                self.log.watch_full("ep_length", l, history_length=100)

    def generate_rollout(self):
        """Fill the rollout buffers with N steps from A envs (rl/rollout.py:702-969).

        With a `SplitVecEnv` of array-stepping parts the step is software-pipelined: the GPU runs the policy
        for one group of envs while the host steps the other group, so neither waits for the whole of the
        other (the reference overlaps nothing: rl/rollout.py:792-816 is forward -> sync -> step)."""
        assert self.vec_env is not None, "Please attach vector environment first."
        N, A = self.N, self.A
        env = self.vec_env
        parts = getattr(env, "parts", [env])
        if all(hasattr(p, "step_arrays") for p in parts):
            self._rollout_pipelined(parts)
        else:
            self._rollout_generic(env)
        self.ext_rewards.copy_(self._rewards_host, non_blocking=True)
        self.terminals.view(torch.uint8).copy_(self._dones_host, non_blocking=True)
        self._sample_calls += N + 1
        self.step += N * A * self.world

    def _rollout_pipelined(self, parts):
        N = self.N
        rew_np, done_np = self._rewards_host.numpy(), self._dones_host.numpy()
        act_np = self._actions_host.numpy()
        bounds = np.cumsum([0] + [p.num_envs for p in parts]).tolist()
        assert bounds[-1] == self.A
        P = len(parts)
        if len(self._step_events) < P:
            self._step_events = [torch.cuda.Event() for _ in range(P)]
        events = self._step_events

        def enqueue(i, t):
            # H2D of group i's observations (pinned -> HBM), policy + sampling, actions D2H; all async
            lo, hi = bounds[i], bounds[i + 1]
            self.all_obs[t, lo:hi].copy_(parts[i].obs_t, non_blocking=True)
            self._policy_step(t, lo, hi)
            if t < N:
                self._actions_host[lo:hi].copy_(self.actions[t, lo:hi], non_blocking=True)
                events[i].record()

        def step_envs(i, t):
            # the one host wait of group i's step: its actions have landed; then step it on host cores
            lo, hi = bounds[i], bounds[i + 1]
            events[i].synchronize()
            parts[i].step_arrays(act_np[lo:hi], rew_np[t, lo:hi], done_np[t, lo:hi])
            _, ep_len, ep_score = parts[i].last_episode_stats
            self._log_finished(done_np[t, lo:hi].astype(bool), ep_len, ep_score)

        for t in range(N + 1):
            for i in range(P):
                enqueue(i, t)  # needs obs(i, t): group i was stepped for t-1 below
                j, tj = (i + 1) % P, (t if i + 1 == P else t - 1)
                if 0 <= tj < N:
                    step_envs(j, tj)  # overlaps the GPU work just queued for group i
        self.obs = parts[0].obs if P == 1 else np.concatenate([p.obs for p in parts])

    def _rollout_generic(self, env):
        """gym-API vector env (`step(actions) -> obs, rew, done, infos`), one group."""
        N = self.N
        rew_np, done_np = self._rewards_host.numpy(), self._dones_host.numpy()
        act_np = self._actions_host.numpy()
        stream = torch.cuda.current_stream()
        for t in range(N + 1):
            self.all_obs[t].copy_(torch.from_numpy(np.ascontiguousarray(self.obs)), non_blocking=True)
            self._policy_step(t)
            if t == N:
                break  # final state: only its value estimate is needed (rl/rollout.py:871-878)
            self._actions_host.copy_(self.actions[t], non_blocking=True)
            stream.synchronize()  # the one device->host sync of the step: the envs need the actions
            self.obs, rew, dones, infos = env.step(act_np.copy())
            rew_np[t] = rew
            done_np[t] = dones
            self._log_finished(np.asarray(dones, bool), np.asarray([i.get("ep_length", 0) for i in infos]),
                               np.asarray([i.get("ep_score", 0.0) for i in infos]))

    # ------------------------------------------------------------------ returns
    def calculate_returns(self):
        """Advantages (lambda_policy) and value targets (lambda_value) in one fused scan
        (rl/rollout.py:1182-1285 -> rl/returns.py:7-67)."""
        N, A = self.N, self.A
        assert self.VH == 1, "one extrinsic value head on this path"
        value = self.value.view(N + 1, A)
        self._call("ppo_gae_scan_f32", _p(self.ext_rewards), _p(value), _p(value[N]), _p(self.terminals),
                   _lib.PPO_TERM_U8, _p(self.advantage), _p(self.returns), N, A, A, float(args.gamma),
                   float(args.lambda_policy), float(args.lambda_value), _lib.PPO_SCAN_AUTO)

    # ------------------------------------------------------------------ training
    def _normalize_advantages(self):
        """(a - mean) / (std + eps) over the whole (global) batch (rl/rollout.py:1887-1900)."""
        n = self.N * self.A
        self._call("ppo_moments_f64", _p(self.advantage), n, _p(self._moments), _p(self._moments_ws))
        parallel.allreduce_sum_(self._moments)
        self._call("ppo_normalize_f32", _p(self.advantage), n, _p(self._moments), float(args.advantage_epsilon),
                   _p(self.norm_advantage), _p(self._mean_std))

    def optimizer_step(self, label="policy"):
        """All-reduce (DP) + global-norm clip + Adam in the flat buffer (rl/rollout.py:1287-1321)."""
        net, cfg = self.net, args.policy_opt
        parallel.allreduce_sum_(net.grad)
        net.adam_step(lr=cfg.lr, beta1=cfg.adam_beta1, beta2=cfg.adam_beta2, eps=cfg.adam_epsilon,
                      max_grad_norm=args.max_grad_norm if args.grad_clip_mode == "global_norm" else 0.0,
                      grad_div=float(self.world), grad_norm_out=self._grad_norm)

    def train_policy(self):
        """PPO epochs over the rollout (rl/rollout.py:1853-1953, 2257-2407)."""
        N, A = self.N, self.A
        B = N * A
        net = self.net
        cfg = args.policy_opt
        mb = parallel.local_minibatch(cfg.mini_batch_size)  # the flag is the GLOBAL minibatch
        if B % mb:
            raise ValueError(f"batch {B} is not a multiple of the per-rank minibatch {mb}")
        n_mb = B // mb
        self._normalize_advantages()
        obs_rows = self.all_obs[:N].view(B, -1)
        row_bytes = obs_rows.shape[1] * obs_rows.element_size()
        mb_obs = net._buf("mb_obs", (mb, *self.state_shape), self.all_obs.dtype)
        stat_rows = net._buf("stat_rows", (cfg.epochs * n_mb, 8))
        norm_rows = net._buf("norm_rows", (cfg.epochs * n_mb,))
        k = 0
        for epoch in range(cfg.epochs):
            ordering = np.arange(B, dtype=np.int32)
            np.random.shuffle(ordering)  # host RNG, as the reference (rl/rollout.py:2319-2320)
            order_dev = torch.from_numpy(ordering).to(self.device, non_blocking=True)
            for j in range(n_mb):
                idx = order_dev[j * mb:(j + 1) * mb]
                self._call("ppo_gather_rows", _p(obs_rows), row_bytes, B, _p(idx), mb, _p(mb_obs))
                stats = net.ppo_minibatch(mb_obs, self.actions, self.log_pac, self.log_policy, self.norm_advantage,
                                          self.returns, eps_clip=self.ppo_epsilon, ent_coef=self.current_entropy_bonus,
                                          vf_coef=args.ppo_vf_coef, loss_scale=1.0, index=idx)
                self.optimizer_step()
                # keep the minibatch statistics on the device: column sums -> one row per minibatch
                self._call("ppo_colsum_f32", _p(stats), mb, 8, 8, _p(stat_rows[k]), 0)
                norm_rows[k:k + 1].copy_(self._grad_norm, non_blocking=True)
                k += 1
        self._stat_rows = (stat_rows, norm_rows, mb)

    def train(self):
        """rl/rollout.py:2220-2255 for the single architecture: policy (+value heads) only."""
        self.train_policy()
        self.batch_counter += 1

    def fetch_stats(self):
        """ONE device->host copy per iteration with everything the reference logs per minibatch
        (rl/rollout.py:1685-1691, 1759-1769, 1317)."""
        if self._stat_rows is None:
            return {}
        stat_rows, norm_rows, mb = self._stat_rows
        s = stat_rows.cpu().numpy().astype(np.float64) / mb
        out = {"loss_pg": s[:, 0].mean(), "entropy": s[:, 1].mean(), "loss_v_ext": s[:, 2].mean(),
               "clip_frac": s[:, 3].mean(), "kl_approx": s[:, 4].mean(), "kl_true": s[:, 5].mean(),
               "loss_policy": s[:, 6].mean(), "grad_policy": float(norm_rows.mean().item()),
               "adv_mean": float(self._mean_std[0].item()), "adv_std": float(self._mean_std[1].item())}
        if not args.disable_logging:
            for k_, v in out.items():
                self.log.watch_mean(k_, v)
        return out

    # ------------------------------------------------------------------ checkpoints (rl/rollout.py:394-517)
    def save_checkpoint(self, filename, step, disable_log=False, disable_replay=False, disable_env_state=False):
        data = {"step": step, "ep_count": self.ep_count, "batch_counter": self.batch_counter,
                "model_state_dict": self.model.state_dict(),
                "policy_optimizer_state_dict": self.net.optimizer_state_dict()}
        torch.save(data, filename)

    def load_checkpoint(self, checkpoint_path):
        cp = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        self.model.load_state_dict(cp["model_state_dict"])
        if "policy_optimizer_state_dict" in cp:
            self.net.load_optimizer_state_dict(cp["policy_optimizer_state_dict"])
        self.step = cp["step"]
        self.ep_count = cp.get("ep_count", 0)
        self.batch_counter = cp.get("batch_counter", 0)
        return self.step
