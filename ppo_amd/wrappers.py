"""Vector-env wrappers that sit either side of the rollout every step — host mirror of the reference's
rl/wrappers.py `VecNormalizeRewardWrapper` (:795-919) and `VecRepeatedActionPenalty` (:758-793).

No gym in this image: a wrapper here is a plain object that forwards unknown attributes to the env it
wraps (`num_envs`, `seed`, `save_state`...), which is all the trainer relies on.
"""
import math

import numpy as np

from .running_stats import RunningMeanStd


class VecWrapper:
    # The group-stepping interface of the Runner's pipelined rollout is NOT inherited from the wrapped env: a wrapper
    # that only overrides step() must see every step, so unless it defines `parts` itself (and does its per-step work in
    # `_after_group_step` / `finish_rollout`, as the two wrappers below do) the Runner falls back to stepping it whole.
    _NOT_FORWARDED = frozenset({"env", "parts", "finish_rollout", "step_arrays", "step_upload", "leaves", "obs_t"})

    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):  # only called for attributes not found on the wrapper itself
        if name in VecWrapper._NOT_FORWARDED:
            raise AttributeError(name)
        return getattr(self.env, name)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def step(self, actions):
        return self.env.step(actions)

    # A rollout can be redone from a saved state (Runner.generate_rollout after a failed split launch) when every layer
    # can be put back exactly: a wrapper's own state is its attributes (arrays and numbers, copied), the wrapped env
    # answers for itself.  Only wrappers that say so take part (the two below); any other falls back to "go on from here".
    _snapshot_exact = False

    @property
    def exact_snapshot(self):
        return self._snapshot_exact and bool(getattr(self.env, "exact_snapshot", False))

    def snapshot_state(self):
        import copy
        return copy.deepcopy({k: v for k, v in self.__dict__.items() if k not in ("env", "_parts")})

    def restore_snapshot(self, state):
        import copy
        self.__dict__.update(copy.deepcopy(state))

    def close(self):
        return self.env.close()


class _Part:
    """One array-stepping group of a wrapped vector env (the Runner's pipelined rollout steps groups, not the whole
    env): forwards to the inner env's group; a wrapper's own per-group work goes into `after_step`."""

    def __init__(self, wrapper, inner, lo):
        self.wrapper, self.inner, self.lo = wrapper, inner, lo
        self.hi = lo + inner.num_envs
        self.num_envs = inner.num_envs

    def __getattr__(self, name):
        if name == "inner":
            raise AttributeError(name)
        return getattr(self.inner, name)

    def step_arrays(self, actions, rew_out=None, done_out=None):
        out = self.inner.step_arrays(actions, rew_out, done_out)
        self.wrapper._after_group_step(self, actions)
        return out


def _inner_parts(env):
    parts = getattr(env, "parts", None)
    if parts is None or not all(hasattr(p, "step_arrays") for p in parts):
        return None
    return parts


def get_wrapper(env, wrapper_type):
    """Walk the `.env` chain for the first wrapper of the given type (rl/wrappers.py `get_wrapper`)."""
    while env is not None:
        if isinstance(env, wrapper_type):
            return env
        env = env.__dict__.get("env")
    return None


class VecRepeatedActionPenalty(VecWrapper):
    """Subtract `penalty` from the reward of any env that has repeated one action more than
    `max_repeated_actions` times in a row; action -1 (env skipped) neither counts nor resets."""

    _snapshot_exact = True

    def __init__(self, env, max_repeated_actions: int, penalty: float = 1):
        super().__init__(env)
        self.max_repeated_actions = max_repeated_actions
        self.penalty = penalty
        self.prev_actions = np.zeros([env.num_envs], dtype=np.int32)
        self.duplicate_counter = np.zeros([env.num_envs], dtype=np.int32)

    def reset(self, **kwargs):
        self.prev_actions[:] = 0
        self.duplicate_counter[:] = 0
        return self.env.reset()

    def step(self, actions):
        obs, rewards, dones, infos = self.env.step(actions)
        actions = np.asarray(actions)
        repeated = (actions == self.prev_actions) & (actions >= 0)
        self.duplicate_counter = (self.duplicate_counter + repeated) * repeated
        over = self.duplicate_counter > self.max_repeated_actions
        infos[0]["max_repeats"] = np.max(self.duplicate_counter)
        infos[0]["mean_repeats"] = np.mean(self.duplicate_counter)
        for i in np.flatnonzero(over):
            infos[i]["repeated_action"] = actions[i]
        self.prev_actions[:] = actions
        return obs, rewards - (over * self.penalty), dones, infos

    # ---- group-wise stepping: the counters are per env, so a group updates its own slice; the penalties of a rollout
    # are kept and subtracted in finish_rollout, AFTER the inner wrappers have had the rewards (the reference's order:
    # the penalty comes off the normalised reward)
    @property
    def parts(self):
        if "_parts" not in self.__dict__:
            inner = _inner_parts(self.env)
            lo, parts = 0, []
            for q in inner or []:
                parts.append(_Part(self, q, lo))
                lo += q.num_envs
            self._parts = parts if inner else [self]
            self._penalties = [[] for _ in parts]
        return self._parts

    def _after_group_step(self, part, actions):
        actions = np.asarray(actions)
        sl = slice(part.lo, part.hi)
        repeated = (actions == self.prev_actions[sl]) & (actions >= 0)
        self.duplicate_counter[sl] = (self.duplicate_counter[sl] + repeated) * repeated
        over = self.duplicate_counter[sl] > self.max_repeated_actions
        self.prev_actions[sl] = actions
        self._penalties[self._parts.index(part)].append(over * self.penalty)

    def finish_rollout(self, rewards, dones):
        """rewards / dones [N, A] of the rollout just stepped group by group, in place."""
        if hasattr(self.env, "finish_rollout"):
            self.env.finish_rollout(rewards, dones)
        for part, rows in zip(self._parts, self._penalties):
            if rows:
                rewards[:len(rows), part.lo:part.hi] -= np.stack(rows).astype(rewards.dtype)
            rows.clear()


class VecNormalizeRewardWrapper(VecWrapper):
    """Scale rewards so discounted returns have roughly unit variance, then clip:
        R <- r + gamma * R * (1 - done);  running var over every R seen;  r' = clip(r / sqrt(var + 1e-2), +-clip) * scale

    `moments_sync`, if given, maps np.array([sum, sum_sq, n]) of this step's R to the same moments summed
    over all data-parallel ranks, so every rank keeps the normaliser a single process with all envs would.

    Attribute names (`ret_rms`, `ret_var`, `current_returns`, `std`, `mean`) and the save_state keys are the
    reference's (rl/wrappers.py:795-919): checkpoints and the trainer's `reward_scale` read them."""

    _snapshot_exact = True
    MODES = ("rms", "ema", "custom")
    STATE_KEYS = ("ret_rms", "ret_var", "current_returns")
    epsilon = 1e-2  # added to the variance under the square root

    def __init__(self, env, initial_state=None, gamma: float = 1.0, clip: float = 10.0, scale: float = 1.0,
                 returns_transform=lambda x: x, mode: str = "rms", ed_type=None, ed_bias: float = 1.0,
                 ema_horizon: float = 5e6, moments_sync=None):
        if ed_type is not None:
            raise NotImplementedError("episodic discounting normalisation is outside the PPO hot path")
        if mode not in self.MODES:
            raise ValueError(f"Invalid mode {mode}")
        super().__init__(env)
        # options
        self.mode, self.gamma, self.clip, self.scale = mode, gamma, clip, scale
        self.returns_transform, self.ema_horizon, self.moments_sync = returns_transform, ema_horizon, moments_sync
        # state: per-env discounted return so far, running moments of all of them, EMA variance (mode "ema")
        self.current_returns = np.zeros([env.num_envs], dtype=np.float32)
        self.ret_rms = RunningMeanStd(shape=())
        self.ret_var = 0.0
        if initial_state is not None:
            self.ret_rms.restore_state(initial_state)

    def reset(self):
        self.current_returns *= 0
        return self.env.reset()

    def _update(self, x):
        if self.moments_sync is None:
            self.ret_rms.update(x)
            return
        x64 = np.asarray(x, np.float64)
        s, ss, n = self.moments_sync(np.array([x64.sum(), np.square(x64).sum(), float(x64.shape[0])]))
        mean = s / n
        self.ret_rms.update_from_moments(mean, max(ss / n - mean * mean, 0.0), n)

    def _track(self, rewards, dones):
        """Advance the per-env discounted returns by one step and fold them into the running statistics."""
        self.current_returns = rewards + self.gamma * self.current_returns * (1 - dones)
        self._update(self.returns_transform(self.current_returns))
        if self.mode == "ema":
            alpha = 1 - (len(dones) / min(self.ret_rms.count, self.ema_horizon))
            self.ret_var = alpha * self.ret_var + (1 - alpha) * np.var(self.current_returns)

    def step(self, actions):
        obs, rewards, dones, infos = self.env.step(actions)
        self._track(rewards, dones)
        out = rewards / self.std
        if self.clip is not None and self.clip >= 0:
            bounded = np.clip(out, -self.clip, +self.clip)
            n_clipped = np.sum(bounded != out)
            if n_clipped > 0:
                infos[0]["reward_clips"] = n_clipped
            out = bounded
        return obs, out * self.scale, dones, infos

    # ---- group-wise stepping: the statistics of step t take in EVERY env's return of step t before any reward of that
    # step is scaled, so the groups only record; finish_rollout replays the rollout step by step - the same arithmetic
    # on the same arrays as `step`, hence the same bits
    @property
    def parts(self):
        if "_parts" not in self.__dict__:
            inner = _inner_parts(self.env)
            lo, parts = 0, []
            for q in inner or []:
                parts.append(_Part(self, q, lo))
                lo += q.num_envs
            self._parts = parts if inner else [self]
        return self._parts

    def _after_group_step(self, part, actions):
        pass

    def finish_rollout(self, rewards, dones):
        """rewards / dones [N, A] of the rollout just stepped group by group: normalised in place."""
        if hasattr(self.env, "finish_rollout"):
            self.env.finish_rollout(rewards, dones)
        for t in range(rewards.shape[0]):
            raw = rewards[t].copy()
            self._track(raw, dones[t].astype(bool))
            out = raw / self.std
            if self.clip is not None and self.clip >= 0:
                out = np.clip(out, -self.clip, +self.clip)
            rewards[t] = out * self.scale

    @property
    def mean(self):
        return self.ret_rms.mean

    @property
    def std(self):
        return math.sqrt((self.ret_rms.var if self.mode == "rms" else self.ret_var) + self.epsilon)

    def save_state(self, buffer):
        buffer.update(ret_rms=self.ret_rms.save_state(), ret_var=self.ret_var, current_returns=self.current_returns)

    def restore_state(self, buffer):
        moments, self.ret_var, self.current_returns = (buffer[k] for k in self.STATE_KEYS)
        self.ret_rms.restore_state(moments)
