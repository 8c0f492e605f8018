"""TVF host helpers — mirror of the set-up-time functions of the reference's rl/tvf.py:
`get_value_head_horizons` (:576-610) and `horizon_interpolate` (:527-573).  Both are O(heads) /
O(batch x heads) scalar work done when the runner is built or when values are logged, so they stay on
the host in NumPy ("setup-time / rare"); the per-batch heavy lifting of TVF — the
truncated return estimator — is `ppo_amd.returns_truncated` on the GPU.
"""
import math

import numpy as np

from .returns_truncated import get_return_estimate  # noqa: F401  (rl.tvf callers import it from here too)


def get_value_head_horizons(n_heads: int, max_horizon: int, spacing: str = "geometric", include_weight=False):
    """Horizons spaced (approximately) geometrically from 0 to max_horizon; heads that round to the same
    horizon are merged and their multiplicity returned as the loss weight."""
    if spacing == "linear":
        result = np.asarray(np.round(np.linspace(0, max_horizon, n_heads)), dtype=np.int32)
        return (result, np.ones([n_heads], dtype=np.float32)) if include_weight else result
    if spacing != "geometric":
        raise ValueError(f"Invalid spacing value {spacing}")

    def candidates(count):
        return np.asarray(np.round(np.geomspace(1, max_horizon + 1, count)) - 1, dtype=np.int32)

    count = n_heads
    distinct = len(np.unique(candidates(count)))
    while distinct != n_heads:  # grow in sqrt(n) strides, shrink one at a time (rl/tvf.py:595-599)
        count = count + int(math.sqrt(n_heads)) if distinct < n_heads else count - 1
        distinct = len(np.unique(candidates(count)))
    horizons, multiplicity = np.unique(candidates(count), return_counts=True)
    horizons = horizons.astype(np.int32)
    return (horizons, multiplicity.astype(np.float32)) if include_weight else horizons


def horizon_interpolate(horizons: np.ndarray, values: np.ndarray, target_horizons: np.ndarray):
    """Linear interpolation of values[..., K] (given at sorted `horizons`, horizons[0] == 0) at one target
    horizon per example; targets are clamped to the horizon range and h <= 0 has value 0 by definition."""
    horizons = np.asarray(horizons)
    assert len(set(horizons.tolist())) == len(horizons), f"Horizons duplicates not supported {horizons}"
    assert np.all(np.diff(horizons) > 0), f"Horizons must be sorted and unique horizons:{horizons}"
    assert horizons[0] == 0, "first horizon must be 0"
    values = np.asarray(values)
    *shape, K = values.shape
    shape = tuple(shape)
    assert horizons.shape == (K,)
    targets = np.clip(np.asarray(target_horizons), horizons[0], horizons[-1])
    assert targets.shape == shape, f"{targets.shape} != {shape}"
    flat_v = values.reshape(-1, K)
    flat_t = targets.reshape(-1)
    hi = np.searchsorted(horizons, flat_t, side="left")
    lo = np.maximum(hi - 1, 0)
    rows = np.arange(flat_t.shape[0])
    span = (horizons[hi] - horizons[lo]).astype(np.float64)
    span[span == 0] = 1.0  # only at the lower boundary, where the numerator is 0 as well
    frac = (flat_t - horizons[lo]) / span
    result = flat_v[rows, lo] * (1 - frac) + flat_v[rows, hi] * frac
    result[hi == 0] = 0
    return result.reshape(shape)


def get_rediscounted_value_estimate(values, old_gamma: float, new_gamma: float, horizons, clipping=10):
    """Re-discount per-horizon values [B, K] from old_gamma to new_gamma (rl/tvf.py:388-433): the reward mass
    between consecutive horizons is re-weighted by (new/old)^mid_h, ratio clipped; equal gammas return the
    longest horizon.  NumPy in -> NumPy out, tensor in -> tensor out (tiny: setup-time / rare, R8)."""
    import torch
    B, K = values.shape
    if old_gamma == new_gamma:
        return values[:, -1]
    assert K == len(horizons), f"missmatch {K} {horizons}"
    assert horizons[0] == 0, "first horizon must be 0"
    is_numpy = isinstance(values, np.ndarray)
    v = torch.from_numpy(values) if is_numpy else values
    total = torch.zeros([B], dtype=torch.float32, device=v.device)
    prev, prev_h = v[:, 0], 0
    for i, h in enumerate(horizons[1:], start=1):
        mid_h = ((prev_h + 1 + h) / 2) - 1
        ratio = min((new_gamma ** mid_h) / (old_gamma ** mid_h), clipping)
        total += (v[:, i] - prev) * ratio
        prev, prev_h = v[:, i], h
    return total.numpy() if is_numpy else total


def trim_horizons(horizons, tvf_value_estimates, time, timeout: int, method: str = "timelimit", mode: str = "interpolate",
                  trim_clip: float = -1.0, episode_lengths=(), eta_percentile: float = 90, eta_buffer: int = 32,
                  eta_minh: int = 128):
    """Reduce horizons that reach past the end of the episode back to the time that is left (rl/tvf.py:91-208): a value
    at horizon h equals the value at min(h, time till termination).

    tvf_value_estimates [A, K, VH] (one env step), time [A] (steps since episode start of the states behind them).
    Returns (trimmed [A, K, VH] float32, final_value [A] float32 = mean over the valid horizons of the UNTRIMMED ext
    estimates, time_till_termination [A]) — or (estimates with h = 0 zeroed, 0, None) for method "off".  Host NumPy,
    vectorised over envs; every float operation happens in the reference's type and order (pinned bit for bit by
    tests/golden/tvf_trim_golden.npz).  Mode "random" draws from np.random in the reference's order."""
    horizons = np.asarray(horizons)
    old = np.array(tvf_value_estimates, copy=True)
    new = np.array(tvf_value_estimates, copy=True)
    assert horizons[0] == 0, "First horizon must be zero"
    old[:, 0, :] = 0  # by definition h = 0 is 0.0
    new[:, 0, :] = 0
    if method == "off":
        return new, 0, None
    time = np.asarray(time)
    if method == "timelimit":
        ttt = np.maximum(timeout - time, 0)
    elif method == "est_term":
        est_ep_length = np.percentile(list(episode_lengths) + list(time), eta_percentile).astype(int) + eta_buffer
        est_ep_length += eta_minh / 4  # apply small buffer
        ttt = np.minimum(np.maximum(timeout - time, 0), np.maximum(est_ep_length - time, eta_minh))
    else:
        raise ValueError(f"Invalid trimming method {method}")
    A, K, VH = new.shape
    trimmed_ks = np.searchsorted(horizons, ttt)
    ks = np.arange(K)
    if mode == "interpolate":
        def log_scale(x):
            return np.log10(10 + x) - 1
        at_ttt = horizon_interpolate(log_scale(horizons), old[..., 0], log_scale(ttt))
        past = ks[None, :] >= trimmed_ks[:, None]  # every horizon past the end takes the value interpolated at the end
        new[..., 0] = np.where(past, at_ttt[:, None].astype(new.dtype), new[..., 0])
    elif mode == "average":
        # running mean of the untrimmed estimates from the first trimmed horizon on (sequential float32 sums, k ascending)
        acc = np.zeros((A, VH), dtype=old.dtype)
        count = np.zeros((A, 1), dtype=old.dtype)
        trims = trimmed_ks < K - 1  # otherwise nothing is trimmed
        for k in range(K):
            sel = trims & (k >= trimmed_ks)
            if sel.any():
                acc[sel] += old[sel, k, :]
                count[sel] += 1
                new[sel, k, :] = acc[sel] / count[sel]
    elif mode == "substitute":
        # the shortest horizon that covers the remaining time; like the reference this keeps the ext column only
        new = np.take_along_axis(new[:, :, 0], np.minimum(trimmed_ks[:, None], ks[None, :]), axis=1)[:, :, None]
    elif mode == "random":
        for a, trimmed_k in zip(range(A), trimmed_ks):
            new_ks = np.arange(K)
            for k in range(trimmed_k, K):
                new_ks[k] = np.random.randint(trimmed_k, k + 1)
            new[a, range(K)] = old[a, new_ks]
    else:
        raise ValueError(f"Invalid trimming mode {mode}")
    # the value used for advantages: mean of the untrimmed ext estimates over all valid horizons (at least one)
    final = np.zeros([A], dtype=np.float32)
    first = np.minimum(trimmed_ks, K - 1)
    for k in np.unique(first):  # envs that share a first valid horizon are averaged together (row-wise pairwise sums)
        rows = np.nonzero(first == k)[0]
        final[rows] = np.ascontiguousarray(old[rows, k:, 0]).mean(axis=1)
    if trim_clip >= 0:
        new = old + np.clip(new - old, -trim_clip, +trim_clip)
    return new, final, ttt


class TVFRunnerModule:
    """Rollout-side TVF state (rl/tvf.py:18-386) with the buffers in HBM: per-horizon value estimates
    `tvf_value [N+1, A, K, VH]` written by the rollout's policy step and `tvf_returns [N, A, K, VH]` filled
    by calculate_tvf_returns from the HIP truncated-returns kernel.  With `--tvf_trimming` the policy step records
    into `tvf_untrimmed_value`, and `apply_trimming` (once per rollout, host) fills `tvf_value` and `tvf_final_value`
    step by step exactly as the reference's per-step call does (rl/rollout.py:788-804, 884-892)."""

    def __init__(self, parent):
        import collections

        import torch
        from .config import args
        if args.tvf.trimming not in ("off", "timelimit", "est_term"):
            raise ValueError(f"Invalid trimming method {args.tvf.trimming}")
        if args.tvf.trimming_mode not in ("interpolate", "average", "substitute", "random"):
            raise ValueError(f"Invalid trimming mode {args.tvf.trimming_mode}")
        if args.tvf.trim_advantages not in ("trimmed", "untrimmed", "average"):
            raise ValueError(f"Invalid advantage trimming mode {args.tvf.trim_advantages}.")
        if not 0 <= args.tvf.horizon_dropout < 1:
            raise ValueError("tvf_horizon_dropout must be in [0, 1)")
        if args.tvf.head_weighting not in ("off", "h_weighted"):
            raise ValueError(f"Invalid head weighting {args.tvf.head_weighting}")
        self.runner = parent
        self.head_weighting = args.tvf.head_weighting
        self.trimming = args.tvf.trimming != "off"
        if self.trimming and args.env.timeout <= 0:
            raise ValueError("--tvf_trimming needs the episode step limit: set --env_timeout")
        N, A, K, VH = parent.N, parent.A, len(parent.tvf_horizons), parent.VH
        dev = parent.device
        self.tvf_value = torch.zeros((N + 1, A, K, VH), dtype=torch.float32, device=dev)
        self.tvf_untrimmed_value = torch.zeros_like(self.tvf_value) if self.trimming else self.tvf_value
        self.tvf_final_value = torch.zeros((N + 1, A), dtype=torch.float32, device=dev)
        self.tvf_returns = torch.zeros((N, A, K, VH), dtype=torch.float32, device=dev)
        self.episode_length_buffer = collections.deque([1000], maxlen=1000)  # rl/rollout.py:292, 553-555
        self._dropout_calls = 0

    def trim_horizons(self, tvf_value_estimates, time, method: str = "timelimit", mode: str = "interpolate"):
        """The reference's method signature (rl/tvf.py:91) over the module-level function."""
        from .config import args
        return trim_horizons(self.runner.tvf_horizons, tvf_value_estimates, time, args.env.timeout, method, mode,
                             trim_clip=args.tvf.trim_clip, episode_lengths=self.episode_length_buffer,
                             eta_percentile=args.tvf.eta_percentile, eta_buffer=args.tvf.eta_buffer,
                             eta_minh=args.tvf.eta_minh)

    def apply_trimming(self, all_time, finished_lengths):
        """Trim a whole rollout: all_time [N+1, A] (env time of every recorded state), finished_lengths[t] = lengths of
        the episodes that ended at env step t.  One device->host copy of the untrimmed estimates, the reference's
        per-step call for t = 0..N with the episode-length buffer growing as the rollout did (a step's finished
        episodes are appended after its trimming, rl/rollout.py:793 then :833), one copy back."""
        import torch
        from .config import args
        if not self.trimming:
            return
        untrimmed = self.tvf_untrimmed_value.cpu().numpy()
        N1, A, K, VH = untrimmed.shape
        trimmed = np.empty_like(untrimmed)
        final = np.zeros((N1, A), np.float32)
        for t in range(N1):
            tv, fv, _ttt = self.trim_horizons(untrimmed[t], all_time[t], method=args.tvf.trimming,
                                              mode=args.tvf.trimming_mode)
            trimmed[t] = tv  # mode "substitute" returns the ext column only: broadcast over VH like the reference's store
            final[t] = fv
            if t < len(finished_lengths):
                self.episode_length_buffer.extend(finished_lengths[t])
        self.tvf_value.copy_(torch.from_numpy(trimmed))
        self.tvf_final_value.copy_(torch.from_numpy(final))

    def next_dropout_offset(self, n_terms: int) -> int:
        """Counter range of one minibatch's horizon-dropout draws (B * K of them), never reused within a run."""
        off = self._dropout_calls
        self._dropout_calls += int(n_terms)
        return off

    def value_loss_weights(self):
        """Per-head weights of the value-phase TVF loss (rl/tvf.py:51-62): the duplicate-horizon weights, times —
        with `--tvf_head_weighting=h_weighted` — 1 + (max_horizon - h) / return_n_step scaled by
        2 / (min + max) so the loss keeps its magnitude.  float32 arithmetic as in the reference."""
        import numpy as np
        from .config import args
        w = np.asarray(self.runner.tvf_weights, np.float32)
        if self.head_weighting == "h_weighted":
            hw = np.asarray([1 + ((args.tvf.max_horizon - h) / args.tvf_return_n_step) for h in self.runner.tvf_horizons],
                            dtype=np.float32)
            adjustment = 2 / (np.min(hw) + np.max(hw))
            w = (w * hw * adjustment).astype(np.float32)
        return w
    def calculate_tvf_returns(self, value_head: str = "ext", obs=None, rewards=None, dones=None, tvf_return_mode=None,
                              tvf_return_distribution=None, tvf_n_step=None):
        """[N, A, K] truncated return estimates for the rollout (rl/tvf.py:210-271)."""
        from .config import args
        r = self.runner
        horizons = np.asarray(r.tvf_horizons)
        values = self.tvf_value[..., r.value_heads.index(value_head)]
        return get_return_estimate(
            mode=tvf_return_mode or args.tvf.return_mode,
            distribution=tvf_return_distribution or args.tvf.return_distribution,
            gamma=args.tvf.gamma,
            rewards=rewards if rewards is not None else r.ext_rewards,
            dones=dones if dones is not None else r.terminals,
            required_horizons=horizons, value_sample_horizons=horizons, value_samples=values.contiguous(),
            n_step=tvf_n_step or args.tvf_return_n_step, max_samples=args.tvf.return_samples,
            use_log_interpolation=args.tvf.return_use_log_interpolation)

    def log_tvf_curve_quality(self, ext_values, ext_targets):
        """Explained variance of the truncated-value curve against fixed n-step Monte-Carlo targets, and of the
        ext value head against bootstrapped returns (rl/tvf.py:274-301)."""
        from . import value_quality
        from .config import args
        r = self.runner
        targets = self.calculate_tvf_returns(value_head="ext", tvf_return_distribution="fixed", tvf_n_step=args.n_steps)
        estimates = self.tvf_value[:r.N, :, :, 0]
        value_quality.log_curve_quality(r.log, estimates, targets, r.tvf_horizons)
        ev = value_quality.explained_variance(value_quality._host(ext_values).ravel(),
                                              value_quality._host(ext_targets).ravel())
        r.log.watch_mean("*ev_ext", ev, history_length=1)

    def get_tvf_ext_value_estimate(self, new_gamma: float):
        """[N+1, A] value estimate for GAE: the longest horizon (rl/tvf.py:304-338), re-discounted when the
        policy gamma differs from the TVF gamma (:353-360)."""
        from .config import args
        r = self.runner
        head = r.value_heads.index("ext")
        trimmed, untrimmed = self.tvf_value[:, :, :, head], self.tvf_untrimmed_value[:, :, :, head]
        how = args.tvf.trim_advantages  # trimmed | untrimmed | average (rl/tvf.py:329-340)
        if abs(new_gamma - args.tvf.gamma) < 1e-8:
            if how == "average":
                if not self.trimming:
                    raise ValueError("--tvf_trim_advantages=average needs --tvf_trimming")
                return self.tvf_final_value  # mean over the valid horizons, stored while trimming
            return (trimmed if how == "trimmed" else untrimmed)[:, :, -1]
        assert how != "average", "Average advantage trimming not supported with rediscounting."
        v = trimmed if how == "trimmed" else untrimmed
        N1, A, K = v.shape
        return get_rediscounted_value_estimate(v.reshape(N1 * A, K), args.tvf.gamma, new_gamma,
                                               r.tvf_horizons).reshape(N1, A)
