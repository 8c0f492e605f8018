"""oracle/returns_truncated.py — TEST INFRASTRUCTURE, NOT PRODUCT.

NumPy restatement of the reference's truncated-horizon (TVF) return estimator, the sampled
weighted-n-step average with horizon interpolation:
  rl/returns_truncated.py:10-139   get_return_estimate       (distribution / mode -> n-step samples [K, C])
  rl/returns_truncated.py:142-174  _interpolate              (linear interpolation between value horizons)
  rl/returns_truncated.py:558-620  _n_step_estimate          (one horizon: mean over its C samples)
  rl/returns_truncated.py:623-693  _calculate_sampled_return_multi_fast
and of rl/tvf.py:527-573 horizon_interpolate, :576-610 get_value_head_horizons.

Definition restated (t = time, a = env, h = required horizon k, n = an n-step sample clipped to h):
  D_i[t] = prod_{j<i, t+j<N} gamma * (1 - done[t+j])          S_n[t] = sum_{i<n, t+i<N} r[t+i] * D_i[t]
  boot[t] = interp(V[t+n], h-n)            for t <  N-n
          = interp(V[N],  h-(N-t))         for t >= N-n      (the rollout ends before n steps are taken)
  ret[t, a, k] = mean_c ( S_n[t] + boot[t] * D_n[t] ),   0 for h == 0.
NumPy's dtypes are kept as in the reference (float32 arrays, float64 for `gamma * (1 - bool)` before it
is rounded back into the float32 discount array), so results agree to float32 round-off.

Parity: PINNED by tests/golden/tvf_golden.npz (outputs of the imported reference, incl. its slow
`_calculate_sampled_return_multi_reference`) and the reference's own test recipe and known answer
(tests/test_tvf.py:10-129): tests/test_oracle_tvf.py.
"""
import math

import numpy as np


def interpolation_plan(sample_horizons, target, use_log=False):
    """How the reference's _interpolate evaluates `target`: ('zero',) | ('exact', i) | ('lerp', i0, i1, f).
    `sample_horizons` strictly ascending.  With use_log the axis is log10(10 + h) - 1 (:565,573-574)."""
    if use_log:
        axis = np.log10(10 + np.asarray(sample_horizons)) - 1
        x = np.log10(10 + target) - 1
        nonpositive = x <= 0
    else:
        axis = np.asarray(sample_horizons)
        x = target
        nonpositive = x <= 0
    if nonpositive:
        return ("zero",)
    idx = int(np.searchsorted(axis, x))
    if idx >= len(axis):
        raise IndexError("target horizon beyond the largest value horizon (the reference raises here too)")
    if axis[idx] == x or idx == 0:
        return ("exact", idx)
    dx = axis[idx] - axis[idx - 1]
    if dx == 0:
        return ("exact", idx - 1)
    return ("lerp", idx - 1, idx, (x - axis[idx - 1]) / dx)


def _apply_plan(plan, values):
    """values [..., V] float32 -> [...] float32, with the reference's arithmetic (python-float weights)."""
    if plan[0] == "zero":
        return values[..., 0] * 0
    if plan[0] == "exact":
        return values[..., plan[1]].copy()
    _, i0, i1, f = plan
    return values[..., i0] * (1 - f) + values[..., i1] * f


def sampled_returns(gamma, rewards, dones, required_horizons, value_sample_horizons, value_samples,
                    n_step_samples, use_log_interpolation=False):
    """[N, A, K] float32 for an explicit sample matrix n_step_samples [K, C] (rl/returns_truncated.py:623-693)."""
    rewards = np.asarray(rewards)
    N, A = rewards.shape
    K = len(required_horizons)
    n_step_samples = np.asarray(n_step_samples)
    assert n_step_samples.shape[0] == K
    C = n_step_samples.shape[1]
    # running discounted reward sum and discount for every prefix length that is used
    needed = set(int(x) for x in n_step_samples.ravel()) | set(int(h) for h in required_horizons)
    S = np.zeros((N, A), np.float32)
    D = np.ones((N, A), np.float32)
    S_of, D_of = {}, {}
    for i in range(int(n_step_samples.max())):
        S[:N - i] += rewards[i:] * D[:N - i]
        D[:N - i] *= gamma * (1 - dones[i:])
        if i + 1 in needed:
            S_of[i + 1], D_of[i + 1] = S.copy(), D.copy()
    out = np.zeros((N, A, K), np.float32)
    for k, h in enumerate(required_horizons):
        h = int(h)
        if h == 0:
            continue
        total = np.zeros((N, A), np.float32)
        boot = np.zeros((N, A), np.float32)
        for n in n_step_samples[k]:
            n = min(int(n), h)
            assert 1 <= n <= N
            boot *= 0
            if h - n > 0:
                boot[:N - n] = _apply_plan(interpolation_plan(value_sample_horizons, h - n, use_log_interpolation),
                                           value_samples[n:-1])
            for i in range(n):  # the rollout ends first: bootstrap from the final state
                boot[N - i - 1] = _apply_plan(interpolation_plan(value_sample_horizons, h - i - 1, use_log_interpolation),
                                              value_samples[-1])
            total += S_of[n] + boot * D_of[n]
        total *= 1 / C
        out[:, :, k] = total
    return out


def draw_n_step_samples(distribution, mode, N, K, required_horizons, n_step=40, max_samples=40, seed=None):
    """The sample matrix [K, C] get_return_estimate builds (rl/returns_truncated.py:60-129), drawing from the
    global NumPy generator in the same order.  mode 'full' is not a sampling mode (handled by the caller)."""
    if distribution == "fixed":
        return np.zeros([K, 1], dtype=np.int32) + n_step
    lamb = 1 - (1 / n_step)
    fn = {"exponential": lambda x: lamb ** x, "uniform": lambda x: 1, "hyperbolic": lambda x: 1 / x,
          "quadratic": lambda x: 1 / (N + (x * x))}.get(distribution)
    if fn is None:
        raise ValueError(f"Invalid distribution {distribution}")
    weights = np.asarray([fn(n) for n in range(1, N + 1)], dtype=np.float32)
    weights /= np.sum(weights)
    if seed is not None:
        np.random.seed(seed)
    support = range(1, len(weights) + 1)
    if mode == "standard":
        s = np.random.choice(support, size=(1, max_samples), replace=True, p=weights)
        return np.repeat(s, K, axis=0)
    if mode == "advanced":
        return np.random.choice(support, size=(K, max_samples), replace=True, p=weights)
    if mode in ("clipped", "adaptive"):
        out = np.zeros([K, max_samples], dtype=np.int32)
        for k in range(K):
            cap = max(required_horizons[k], 1) if mode == "clipped" else max(required_horizons[k] // 2, 1)
            w = weights.copy()
            w[cap:] = 0
            w = w / w.sum()
            out[k, :] = np.random.choice(support, size=max_samples, replace=True, p=w)
        return out
    if mode == "mcx":
        out = np.zeros([K, max_samples], dtype=np.int32)
        for k in range(K):
            if required_horizons[k] <= 2 * n_step:
                out[k, :] = required_horizons[k]
            else:
                out[k, :] = np.random.choice(support, size=max_samples, replace=True, p=weights)
        return out
    raise ValueError(f"Invalid return mode {mode}")


def get_return_estimate(distribution, mode, gamma, rewards, dones, required_horizons, value_sample_horizons,
                        value_samples, n_step=40, max_samples=40, use_log_interpolation=False, seed=None):
    N, A = rewards.shape
    K = len(required_horizons)
    args = (gamma, rewards, dones, required_horizons, value_sample_horizons, value_samples)
    if mode == "full" and distribution != "fixed":
        lamb = 1 - (1 / n_step)
        fn = {"exponential": lambda x: lamb ** x, "uniform": lambda x: 1, "hyperbolic": lambda x: 1 / x,
              "quadratic": lambda x: 1 / (N + (x * x))}.get(distribution)
        if fn is None:
            raise ValueError(f"Invalid distribution {distribution}")
        weights = np.asarray([fn(n) for n in range(1, N + 1)], dtype=np.float32)
        weights /= np.sum(weights)
        if seed is not None:
            np.random.seed(seed)
        out = np.zeros([N, A, K], dtype=np.float32)
        for n, w in zip(range(1, N + 1), weights):
            out += sampled_returns(*args, np.zeros([K, 1], dtype=np.int32) + n, use_log_interpolation) * w
        return out
    samples = draw_n_step_samples(distribution, mode, N, K, required_horizons, n_step, max_samples, seed)
    return sampled_returns(*args, samples, use_log_interpolation)


def horizon_interpolate(horizons, values, target_horizons):
    """rl/tvf.py:527-573: per-example linear interpolation of values[..., K] at target_horizons[...]."""
    horizons = np.asarray(horizons)
    assert horizons[0] == 0 and np.all(np.diff(horizons) > 0)
    shape = values.shape[:-1]
    K = values.shape[-1]
    t = np.clip(np.asarray(target_horizons), horizons[0], horizons[-1]).reshape(-1)
    v = values.reshape(-1, K)
    post = np.searchsorted(horizons, t, side="left")
    pre = np.maximum(post - 1, 0)
    rows = np.arange(len(t))
    dx = (horizons[post] - horizons[pre]).astype(np.float64)
    dx[dx == 0] = 1.0
    f = (t - horizons[pre]) / dx
    out = v[rows, pre] * (1 - f) + v[rows, post] * f
    out[post == 0] = 0
    return out.reshape(shape)


def get_value_head_horizons(n_heads, max_horizon, spacing="geometric", include_weight=False):
    """rl/tvf.py:576-610: (approximately) geometric head horizons with duplicate counts as weights."""
    if spacing == "linear":
        result = np.asarray(np.round(np.linspace(0, max_horizon, n_heads)), dtype=np.int32)
        return (result, np.ones([n_heads], dtype=np.float32)) if include_weight else result
    if spacing != "geometric":
        raise ValueError(f"Invalid spacing value {spacing}")

    def heads(x):
        return np.asarray(np.round(np.geomspace(1, max_horizon + 1, x)) - 1, dtype=np.int32)

    x = n_heads
    while len(set(heads(x).tolist())) != n_heads:
        if len(set(heads(x).tolist())) < n_heads:
            x += int(math.sqrt(n_heads))
        else:
            x -= 1
    vals, counts = np.unique(heads(x), return_counts=True)
    return (vals.astype(np.int32), counts.astype(np.float32)) if include_weight else vals.astype(np.int32)
