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
