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


class TVFRunnerModule:
    """Rollout-side TVF state (rl/tvf.py:18-386) with the buffers in HBM: per-horizon value estimates
    `tvf_value [N+1, A, K, VH]` written by the rollout's policy step and `tvf_returns [N, A, K, VH]` filled
    by calculate_tvf_returns from the HIP truncated-returns kernel.  Horizon trimming (off by default,
    rl/config.py:217) is not built, so trimmed == untrimmed values."""

    def __init__(self, parent):
        import torch
        from .config import args
        if args.tvf.trimming != "off" or args.tvf.horizon_dropout > 0:
            raise NotImplementedError("TVF trimming / horizon dropout are off by default and not built")
        if args.tvf.head_weighting not in ("off", "h_weighted"):
            raise ValueError(f"Invalid head weighting {args.tvf.head_weighting}")
        self.runner = parent
        self.head_weighting = args.tvf.head_weighting
        N, A, K, VH = parent.N, parent.A, len(parent.tvf_horizons), parent.VH
        dev = parent.device
        self.tvf_value = torch.zeros((N + 1, A, K, VH), dtype=torch.float32, device=dev)
        self.tvf_untrimmed_value = self.tvf_value
        self.tvf_returns = torch.zeros((N, A, K, VH), dtype=torch.float32, device=dev)

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

    def get_tvf_ext_value_estimate(self, new_gamma: float):
        """[N+1, A] value estimate for GAE: the longest horizon (rl/tvf.py:304-338), re-discounted when the
        policy gamma differs from the TVF gamma (:353-360)."""
        from .config import args
        r = self.runner
        v = self.tvf_value[:, :, :, r.value_heads.index("ext")]
        if abs(new_gamma - args.tvf.gamma) < 1e-8:
            return v[:, :, -1]
        N1, A, K = v.shape
        return get_rediscounted_value_estimate(v.reshape(N1 * A, K), args.tvf.gamma, new_gamma,
                                               r.tvf_horizons).reshape(N1, A)
