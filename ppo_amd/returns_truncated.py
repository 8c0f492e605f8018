"""Truncated-horizon (TVF) return estimates on the GPU — host mirror of the reference's
rl/returns_truncated.py (`get_return_estimate` :10-139 and `_calculate_sampled_return_multi_fast` :623-693).

Split of work: everything that depends only on the horizon lists and the n-step samples — drawing the
samples from the global NumPy generator exactly as the reference does (same draw order, so a seeded run
picks the same samples), clipping n to the horizon, and resolving every `_interpolate` call
(rl/returns_truncated.py:142-174) into "zero / one column / two columns + float32 weights" — is O(K*C + K*N)
scalar work and stays on the host.  Everything that touches [N, A] data (the running discounted sums and the
bootstrap gathers, 4*(N+1)*A*V + 4*N*A*K bytes) runs in libppo_amd.so (csrc/tvf_returns.hip).

NumPy in -> NumPy out like the reference, or torch GPU tensors in -> torch GPU tensor out.
"""
import numpy as np
import torch

from . import _lib

__all__ = ["get_return_estimate", "calculate_sampled_return_multi", "interpolation_plan", "SampledReturnPlan"]

DISTRIBUTIONS = ("fixed", "exponential", "uniform", "hyperbolic", "quadratic")
MODES = ("standard", "advanced", "clipped", "adaptive", "mcx", "full")


def interpolation_plan(axis, x):
    """(mode, i0, i1, w0, w1) for evaluating the reference's _interpolate(horizons, values, target) with
    `axis` = horizons (already log-transformed if requested) and `x` the (transformed) target:
    mode 0 -> 0, mode 1 -> values[i0], mode 2 -> float32(values[i0]*w0 + values[i1]*w1) with float64 weights:
    the reference's factor is a NumPy float64 scalar, so under NumPy >= 2 promotion the blend is a float64
    expression rounded once into the float32 result."""
    if x <= 0:
        return 0, 0, 0, 0.0, 0.0
    idx = int(np.searchsorted(axis, x))
    if idx >= len(axis):
        # the reference indexes horizons[idx] before its own bounds check and raises here as well
        raise IndexError(f"target horizon {x} is beyond the largest value horizon {axis[-1]}")
    if axis[idx] == x or idx == 0:
        return 1, idx, idx, 0.0, 0.0
    dx = axis[idx] - axis[idx - 1]
    if dx == 0:
        return 1, idx - 1, idx - 1, 0.0, 0.0
    f = (x - axis[idx - 1]) / dx
    return 2, idx - 1, idx, float(1 - f), float(f)


def _weights(distribution, n_step, N):
    lamb = 1 - (1 / n_step)
    if distribution == "exponential":
        w = [lamb ** n for n in range(1, N + 1)]
    elif distribution == "uniform":
        w = [1 for _ in range(1, N + 1)]
    elif distribution == "hyperbolic":
        w = [1 / n for n in range(1, N + 1)]
    elif distribution == "quadratic":
        w = [1 / (N + (n * n)) for n in range(1, N + 1)]
    else:
        raise ValueError(f"Invalid distribution {distribution}")
    w = np.asarray(w, dtype=np.float32)
    w /= np.sum(w)
    return w


def _draw_samples(distribution, mode, N, required_horizons, n_step, max_samples, seed):
    """n-step samples [K, C] in the reference's draw order (rl/returns_truncated.py:65-129)."""
    K = len(required_horizons)
    if distribution == "fixed":
        return np.zeros([K, 1], dtype=np.int32) + n_step
    weights = _weights(distribution, n_step, N)
    if seed is not None:
        np.random.seed(seed)
    support = range(1, len(weights) + 1)
    if mode == "standard":
        return np.repeat(np.random.choice(support, size=(1, max_samples), replace=True, p=weights), K, axis=0)
    if mode == "advanced":
        return np.random.choice(support, size=(K, max_samples), replace=True, p=weights)
    if mode in ("clipped", "adaptive", "mcx"):
        samples = np.zeros([K, max_samples], dtype=np.int32)
        for k in range(K):
            h = required_horizons[k]
            if mode == "mcx":
                if h <= 2 * n_step:
                    samples[k, :] = h
                else:
                    samples[k, :] = np.random.choice(support, size=max_samples, replace=True, p=weights)
                continue
            cap = max(h, 1) if mode == "clipped" else max(h // 2, 1)
            w = weights.copy()
            w[cap:] = 0
            w = w / w.sum()
            samples[k, :] = np.random.choice(support, size=max_samples, replace=True, p=w)
        return samples
    raise ValueError(f"Invalid return mode {mode}")


def _dev(x, dtype):
    dev = torch.device("cuda", torch.cuda.current_device())
    if isinstance(x, torch.Tensor):
        return x.to(device=dev, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x))).to(device=dev, dtype=dtype).contiguous()


class SampledReturnPlan:
    """Everything of one `_calculate_sampled_return_multi_fast` call that does not depend on the [N, A] data, resolved
    on the host and uploaded: the clipped n-step per (k, c), the prefix-cache slots, and every `_interpolate` call as
    (mode, columns, float64 weights).  `launch(rewards, dones, value_samples, out)` is then one C-ABI call
    (bench.py times exactly that)."""

    def __init__(self, N, A, required_horizons, value_sample_horizons, n_step_samples, use_log_interpolation=False,
                 device=None):
        self.lib = _lib.load()
        hz_req = np.asarray(required_horizons).astype(np.int64).reshape(-1)
        hz_val = np.asarray(value_sample_horizons).astype(np.int64).reshape(-1)
        K, V = len(hz_req), len(hz_val)
        samples = np.asarray(n_step_samples).astype(np.int64)
        if samples.ndim != 2 or samples.shape[0] != K:
            raise ValueError(f"n_step_samples must be [K={K}, C], got {samples.shape}")
        C = samples.shape[1]
        live = hz_req > 0
        n_eff = np.minimum(samples, np.maximum(hz_req, 1)[:, None])  # n clipped to the horizon (:590-591)
        if live.any() and (n_eff[live].min() < 1 or n_eff[live].max() > N):
            raise AssertionError("n-step samples must satisfy 1 <= n <= N")  # the reference asserts (:597)
        n_eff = np.clip(n_eff, 1, N)
        max_n = int(n_eff.max())
        used = np.unique(n_eff)
        nd_of_n = np.full(max_n + 1, -1, np.int32)
        nd_of_n[used] = np.arange(len(used), dtype=np.int32)
        nd_index = nd_of_n[n_eff]
        axis = (np.log10(10 + hz_val) - 1) if use_log_interpolation else hz_val

        def plan(target):
            return interpolation_plan(axis, (np.log10(10 + target) - 1) if use_log_interpolation else target)

        main_plan = np.zeros((K, C, 3), np.int32)
        main_w = np.zeros((K, C, 2), np.float64)
        tail_plan = np.zeros((K, N + 1, 3), np.int32)
        tail_w = np.zeros((K, N + 1, 2), np.float64)
        for k in range(K):
            if not live[k]:
                continue
            h = int(hz_req[k])
            for c in range(C):
                m, i0, i1, w0, w1 = plan(h - int(n_eff[k, c]))
                main_plan[k, c] = (m, i0, i1)
                main_w[k, c] = (w0, w1)
            for j in range(1, min(int(n_eff[k].max()), N) + 1):  # only rows t >= N - n ever use the tail
                m, i0, i1, w0, w1 = plan(h - j)
                tail_plan[k, j] = (m, i0, i1)
                tail_w[k, j] = (w0, w1)
        two = main_plan[..., 0] == 2  # the kernel reads a two-column plan as one paired LDS read of (i0, i0 + 1)
        two_t = tail_plan[..., 0] == 2
        assert (main_plan[..., 2][two] == main_plan[..., 1][two] + 1).all() and \
            (tail_plan[..., 2][two_t] == tail_plan[..., 1][two_t] + 1).all(), "two-column plans must name neighbours"
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        self.N, self.A, self.K, self.V, self.C, self.max_n, self.ND = N, A, K, V, C, max_n, len(used)
        self.tables = [up(n_eff.astype(np.int32)), up(nd_index.astype(np.int32)), up(nd_of_n)]
        self.plans = [up(main_plan), up(main_w), up(tail_plan), up(tail_w), up((~live).astype(np.uint8))]
        self.ws_bytes = self.lib.ppo_tvf_returns_workspace_bytes(N, A, self.ND, K, C)
        self.ws = torch.empty((self.ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
        self.device = dev

    def launch(self, gamma, r, d, vs, out):
        p = lambda t: t.data_ptr()  # noqa: E731
        t_n_eff, t_nd_index, t_nd_of_n = self.tables
        t_mp, t_mw, t_tp, t_tw, t_kz = self.plans
        rc = self.lib.ppo_tvf_returns_f32(p(r), p(d), p(vs), self.N, self.A, self.V, self.K, self.C, float(gamma),
                                          p(t_n_eff), p(t_nd_index), p(t_nd_of_n), self.max_n, self.ND, p(t_mp), p(t_mw),
                                          p(t_tp), p(t_tw), p(t_kz), p(self.ws), self.ws_bytes, p(out),
                                          _lib.current_stream())
        _lib.check(rc, "ppo_tvf_returns_f32")
        return out


def calculate_sampled_return_multi(gamma, rewards, dones, required_horizons, value_sample_horizons, value_samples,
                                   n_step_samples, use_log_interpolation=False):
    """[N, A, K] float32 returns for an explicit sample matrix n_step_samples [K, C]
    (the reference's _calculate_sampled_return_multi_fast)."""
    _lib.require_gpu()
    as_numpy = not isinstance(rewards, torch.Tensor)
    r = _dev(rewards, torch.float32)
    N, A = r.shape
    d = _dev(dones, torch.bool).view(torch.uint8)
    vs = _dev(value_samples, torch.float32)
    K, V = len(np.asarray(required_horizons).reshape(-1)), len(np.asarray(value_sample_horizons).reshape(-1))
    if tuple(vs.shape) != (N + 1, A, V) or tuple(d.shape) != (N, A):
        raise ValueError(f"shape mismatch: rewards {tuple(r.shape)}, dones {tuple(d.shape)}, value_samples {tuple(vs.shape)}")
    plan = SampledReturnPlan(N, A, required_horizons, value_sample_horizons, n_step_samples, use_log_interpolation, r.device)
    out = plan.launch(gamma, r, d, vs, torch.empty((N, A, K), dtype=torch.float32, device=r.device))
    return out.cpu().numpy() if as_numpy else out


def get_return_estimate(distribution: str, mode: str, gamma: float, rewards, dones, required_horizons,
                        value_sample_horizons, value_samples, n_step: int = 40, max_samples: int = 40,
                        use_log_interpolation: bool = False, seed=None):
    """Weighted average of sampled n-step truncated returns, [N, A, K] float32
    (reference: rl/returns_truncated.py:10-139; same arguments, defaults and ValueErrors)."""
    if distribution not in DISTRIBUTIONS:
        raise ValueError(f"Invalid distribution {distribution}")
    N, A = rewards.shape
    K = len(required_horizons)
    args = (gamma, rewards, dones, required_horizons, value_sample_horizons, value_samples)
    if distribution != "fixed" and mode not in MODES:
        _weights(distribution, n_step, N)  # same error precedence as the reference
        if seed is not None:
            np.random.seed(seed)
        raise ValueError(f"Invalid return mode {mode}")
    if mode == "full" and distribution != "fixed":
        weights = _weights(distribution, n_step, N)
        if seed is not None:
            np.random.seed(seed)
        total = None
        for n, w in zip(range(1, N + 1), weights):
            part = calculate_sampled_return_multi(*args, np.zeros([K, 1], dtype=np.int32) + n, use_log_interpolation)
            part = part * w  # float32 * numpy float32 scalar, accumulated in sample order like the reference
            total = part if total is None else total + part
        return total
    hz = np.asarray(required_horizons)
    samples = _draw_samples(distribution, mode, N, hz, n_step, max_samples, seed)
    return calculate_sampled_return_multi(*args, samples, use_log_interpolation)
