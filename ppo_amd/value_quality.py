"""Value-quality diagnostics written to the log once per batch / every 4th batch (reference:
rl/rollout.py:971-1110 and 1199-1285, rl/tvf.py:274-301, rl/utils.py:82-104 and 399-414).

Logging only: nothing here feeds the optimiser.  The per-batch moments are taken on the device the rollout
lives on and fetched in one copy; the explained-variance block runs on a quarter of the batches
(batch_counter % 4 == 3) unless --disable_ev, like the reference, because it costs one extra return scan
(and, with TVF, one Monte-Carlo truncated-return pass) plus one forward of A observations.
"""
import numpy as np
import torch

CURVE_MAX_HEADS = 7  # args.sns.max_heads (rl/config.py:195): the curve diagnostics look at <= 7 horizons


def explained_variance(ypred, y, bias: float = 0.0) -> float:
    """1 - Var[y - ypred] / Var[y] clipped to [-1, 1]; nan when y has no variance (rl/utils.py:399-414)."""
    ypred, y = np.asarray(ypred, dtype=np.float64), np.asarray(y, dtype=np.float64)
    if y.ndim != 1 or ypred.ndim != 1:
        raise ValueError("explained_variance takes flat arrays")
    total = float(np.var(y)) + bias
    if total == 0:
        return float("nan")
    return float(np.clip(1.0 - (float(np.var(y - ypred)) + bias) / total, -1.0, 1.0))


def even_sample_down(items, max_values: int):
    """At most max_values of the items, evenly spaced, last one always kept; negative = all (rl/utils.py:82-104)."""
    if type(max_values) is not int:
        raise TypeError("max_values must be an int")
    items = list(items)
    if max_values < 0 or len(items) <= max_values:
        return items
    if max_values == 0:
        return []
    if max_values == 1:
        return items[-1:]
    return [items[i] for i in np.linspace(0, len(items) - 1, max_values, dtype=np.int32)]


def _host(t):
    return t.detach().double().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t, dtype=np.float64)


def log_batch_moments(log, named):
    """watch_mean_std of each (key, tensor, kwargs) with ONE device->host copy for all of them
    (rl/rollout.py:1199, 1252-1256)."""
    named = [(k, t, kw) for k, t, kw in named if t is not None and t.numel()]
    if not named:
        return
    rows = []
    for _, t, _ in named:
        var, mean = torch.var_mean(t.detach().double().flatten(), unbiased=False)
        rows += [mean, var.sqrt()]
    got = torch.stack(rows).cpu().numpy()
    for i, (key, _, kw) in enumerate(named):
        log.watch(f"{key}_mean", float(got[2 * i]), **kw)
        log.watch(f"{key}_std", float(got[2 * i + 1]), **kw)


def log_feature_statistics(log, model_out):
    """min / max / mean / std / sparsity of the encoder features of the first row of the rollout
    (rl/rollout.py:971-983, rl/logger.py:209-229)."""
    for key in ("policy", "value"):
        for which in ("raw_features", "features"):
            f = model_out.get(f"{key}_{which}")
            if f is None:
                continue
            a = _host(f).ravel()
            name = f"*{key}_{which}"
            for suffix, v in (("min", a.min()), ("max", a.max()), ("mean", a.mean()), ("std", a.std()),
                              ("sparsity", 1.0 - np.count_nonzero(a) / a.size)):
                log.watch(f"{name}_{suffix}", float(v))


def log_dna_value_quality(log, values, targets):
    """ev_ext / ev_average and the value / target bias and variance (rl/rollout.py:986-1035); values and targets
    are the [N, A] value estimates and the bootstrapped discounted returns of the same rollout."""
    values, targets = _host(values), _host(targets)
    ev = explained_variance(values.ravel(), targets.ravel())
    log.watch_mean("ev_ext", ev, history_length=1)
    log.watch_mean("ev_average", ev, display_width=8, display_name="ev_avg", history_length=1)
    log.watch_mean("z_value_bias", float(values.mean()), display_width=0, history_length=1)
    log.watch_mean("z_target_bias", float(targets.mean()), display_width=0, history_length=1)
    log.watch_mean("z_value_var", float(values.var()), display_width=0, history_length=1)
    log.watch_mean("z_target_var", float(targets.var()), display_width=0, history_length=1)
    return ev


def log_curve_quality(log, estimates, targets, horizons, postfix: str = "", max_heads: int = CURVE_MAX_HEADS):
    """Explained variance of the truncated-value curve at up to max_heads horizons, plus first / mid / last and the
    variance-weighted average (rl/rollout.py:1038-1110).  estimates, targets: [N, A, K]."""
    estimates, targets = _host(estimates), _host(targets)
    K = estimates.shape[-1]
    est, tgt = estimates.reshape(-1, K), targets.reshape(-1, K)

    def one(k, name=None):
        name = str(k) if name is None else name
        var = float(np.var(tgt[:, k]))
        nev = float(np.var(tgt[:, k] - est[:, k]))
        ev = 0.0 if var == 0 else float(np.clip(1.0 - nev / var, -1.0, 1.0))
        log.watch_mean(f"ev_{name}{postfix}", ev, display_width=0, history_length=1)
        log.watch_mean(f"nev_{name}{postfix}", nev, display_width=0, history_length=1)
        log.watch_mean(f"var_{name}{postfix}", var, display_width=0, history_length=1)
        log.watch_mean(f"*vr_ratio_{name}", float(est[:, k].mean() / (abs(tgt[:, k].mean()) + 1e-6)), history_length=1)
        return var, nev

    # the reference numbers the heads over horizons[start:] but indexes the full arrays with those numbers
    # (rl/rollout.py:1095-1097), so with a leading horizon 0 the last head is never looked at; kept as it is
    start = 1 if int(horizons[0]) == 0 else 0
    heads = even_sample_down(range(K - start), max_heads)
    if not heads:
        return float("nan")
    total_var = total_nev = 0.0
    for k in heads:
        var, nev = one(k)
        total_var += var
        total_nev += nev
    one(heads[0], "first")
    one(heads[-1], "last")
    one(heads[len(heads) // 2], "mid")
    avg = 0.0 if total_var == 0 else float(np.clip(1.0 - total_nev / total_var, -1.0, 1.0))
    log.watch_mean(f"ev_average{postfix}", avg, display_width=8, display_name="ev_avg" + postfix, history_length=1)
    return avg
