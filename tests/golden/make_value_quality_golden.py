#!/usr/bin/env python3
"""Generate tests/golden/value_quality_golden.json by running the REFERENCE's logging helpers on CPU:
utils.explained_variance (rl/utils.py:399-414), utils.even_sample_down (rl/utils.py:82-104),
Runner.log_dna_value_quality's arithmetic (rl/rollout.py:986-1035, given values and targets) and
Runner._log_curve_quality (rl/rollout.py:1038-1110).  Build container only (needs /root/reference; see ref_shim.py):

    python tests/golden/make_value_quality_golden.py

Data only: seeded inputs are re-created by the test from the seeds stored here; outputs are the logged numbers."""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_shim import load_reference  # noqa: E402


class Recorder:
    def __init__(self):
        self.got = {}

    def watch_mean(self, key, value, **kw):
        self.got[key] = float(value)


def curve_inputs(seed, N, A, K, degenerate):
    rng = np.random.default_rng(seed)
    targets = rng.normal(size=(N, A, K)).astype(np.float32).cumsum(axis=2).astype(np.float32)
    estimates = (targets + rng.normal(size=(N, A, K)).astype(np.float32) * np.linspace(0.1, 2.0, K, dtype=np.float32)).astype(np.float32)
    if degenerate:
        targets[..., 0] = 0.0
    return estimates, targets


def main():
    load_reference(["--tvf_enabled=True", "--device=cpu", "--env_reward_normalization=off", "--output_folder=/tmp/ref_golden_out"])
    from rl import rollout, utils
    out = {"ev": [], "sample_down": [], "curve": []}
    rng = np.random.default_rng(5)
    for case in range(6):
        n = 257
        y = rng.normal(size=n).astype(np.float32) * (0 if case == 4 else 1)
        ypred = (y * [1.0, 0.5, -1.0, 0.0, 1.0, 3.0][case] + rng.normal(size=n).astype(np.float32) * 0.3).astype(np.float32)
        bias = 0.25 if case == 5 else 0.0
        ev = utils.explained_variance(ypred, y, bias) if bias else utils.explained_variance(ypred, y)
        out["ev"].append({"y": y.tolist(), "ypred": ypred.tolist(), "bias": bias, "ev": None if np.isnan(ev) else float(ev)})
    for n, m in ((10, 7), (3, 7), (7, 7), (128, 7), (16, 1), (16, 0), (16, -1), (9, 2), (33, 5)):
        out["sample_down"].append({"n": n, "max": m, "got": [int(v) for v in utils.even_sample_down(range(n), m)]})
    for seed, (N, A, K), first_h, degenerate in ((1, (8, 4, 16), 0, False), (2, (8, 4, 5), 1, False), (3, (4, 6, 32), 0, True),
                                                  (4, (4, 3, 1), 1, False)):
        estimates, targets = curve_inputs(seed, N, A, K, degenerate)
        horizons = np.arange(K) + first_h
        for postfix in ("", "_x"):
            log = Recorder()
            me = types.SimpleNamespace(log=log, tvf_horizons=horizons)
            rollout.Runner._log_curve_quality(me, estimates, targets, postfix=postfix)
            out["curve"].append({"seed": seed, "shape": [N, A, K], "first_horizon": first_h, "degenerate": degenerate,
                                 "postfix": postfix, "logged": log.got})
    path = os.path.join(HERE, "value_quality_golden.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
