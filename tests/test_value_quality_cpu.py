"""CPU: the value-quality diagnostics (ppo_amd/value_quality.py) against numbers logged by the reference's own
helpers (tests/golden/value_quality_golden.json, made by tests/golden/make_value_quality_golden.py from
rl/utils.py:82-104, 399-414 and rl/rollout.py:1038-1110)."""
import json
import math
import os

import numpy as np
import pytest

from ppo_amd import value_quality as vq

GOLD = os.path.join(os.path.dirname(__file__), "golden", "value_quality_golden.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


class Recorder:
    def __init__(self):
        self.got, self.kw = {}, {}

    def watch_mean(self, key, value, **kw):
        self.got[key], self.kw[key] = float(value), kw

    def watch(self, key, value, **kw):
        self.got[key] = float(value)


def curve_inputs(seed, N, A, K, degenerate):  # the generator's inputs, re-created from the seed
    rng = np.random.default_rng(seed)
    targets = rng.normal(size=(N, A, K)).astype(np.float32).cumsum(axis=2).astype(np.float32)
    estimates = (targets + rng.normal(size=(N, A, K)).astype(np.float32) * np.linspace(0.1, 2.0, K, dtype=np.float32)).astype(np.float32)
    if degenerate:
        targets[..., 0] = 0.0
    return estimates, targets


def test_explained_variance_matches_reference(gold):
    assert len(gold["ev"]) == 6
    for case in gold["ev"]:
        got = vq.explained_variance(np.asarray(case["ypred"], np.float32), np.asarray(case["y"], np.float32), case["bias"])
        if case["ev"] is None:
            assert math.isnan(got)
        else:
            assert got == pytest.approx(case["ev"], abs=2e-6)  # the reference takes the variances in float32
    with pytest.raises(ValueError):
        vq.explained_variance(np.zeros((2, 2)), np.zeros((2, 2)))


def test_even_sample_down_matches_reference(gold):
    for case in gold["sample_down"]:
        assert vq.even_sample_down(range(case["n"]), case["max"]) == case["got"]
    with pytest.raises(TypeError):
        vq.even_sample_down(range(4), 2.0)


def test_curve_quality_matches_reference(gold):
    assert len(gold["curve"]) == 8
    for case in gold["curve"]:
        N, A, K = case["shape"]
        estimates, targets = curve_inputs(case["seed"], N, A, K, case["degenerate"])
        log = Recorder()
        vq.log_curve_quality(log, estimates, targets, np.arange(K) + case["first_horizon"], postfix=case["postfix"])
        assert set(log.got) == set(case["logged"])
        for key, want in case["logged"].items():
            assert log.got[key] == pytest.approx(want, rel=2e-5, abs=2e-6), key
        assert log.kw["ev_average" + case["postfix"]]["display_name"] == "ev_avg" + case["postfix"]


def test_dna_value_quality_numbers():
    rng = np.random.default_rng(0)
    targets = rng.normal(size=(16, 8))
    values = targets * 0.8 + rng.normal(size=(16, 8)) * 0.1
    log = Recorder()
    ev = vq.log_dna_value_quality(log, values, targets)
    want = 1 - np.var(targets - values) / np.var(targets)
    assert ev == pytest.approx(want, rel=1e-12)
    assert log.got["ev_ext"] == log.got["ev_average"] == ev
    assert log.got["z_value_bias"] == pytest.approx(values.mean())
    assert log.got["z_target_var"] == pytest.approx(targets.var())
    assert set(log.got) == {"ev_ext", "ev_average", "z_value_bias", "z_target_bias", "z_value_var", "z_target_var"}


def test_batch_moments_one_copy():
    torch = pytest.importorskip("torch")
    a, b = torch.arange(12.0).view(3, 4), torch.tensor([1.0, -1.0])
    log = Recorder()
    vq.log_batch_moments(log, [("a", a, {}), ("skip", None, {}), ("b", b, {"display_width": 0})])
    assert log.got == {"a_mean": 5.5, "a_std": pytest.approx(float(np.std(np.arange(12.0)))), "b_mean": 0.0, "b_std": 1.0}
