"""GPU: ppo_amd.returns_truncated (HIP, through the C ABI) against the reference's golden outputs and the
oracle.  Bit-exact: the kernels keep the reference's float32/float64 operation order."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import returns_truncated as T  # noqa: E402 (checker)
from ppo_amd import returns_truncated as R  # noqa: E402


@pytest.fixture(scope="module")
def gold(golden_dir):
    return (np.load(os.path.join(golden_dir, "tvf_golden.npz")),
            json.load(open(os.path.join(golden_dir, "tvf_golden.json"))))


def _base(g, prefix):
    return {k: g[prefix + k] for k in ("rewards", "dones", "required_horizons", "value_sample_horizons", "value_samples")}


def test_reference_test_recipe_bit_exact(gold):
    g, _ = gold
    base = _base(g, "t_")
    K = len(base["required_horizons"])
    for name, samp in (("n1", np.full((K, 1), 1)), ("n8", np.full((K, 1), 8)), ("n128", np.full((K, 1), 128)),
                       ("exp", g["t_samples"])):
        out = R.calculate_sampled_return_multi(0.9997, **base, n_step_samples=samp)
        assert out.dtype == np.float32 and out.shape == g["t_fast_" + name].shape
        assert np.array_equal(out, g["t_fast_" + name]), (name, np.abs(out - g["t_fast_" + name]).max())
        ref = g["t_slow_" + name]  # the reference's own acceptance test (tests/test_tvf.py:46)
        if name != "n128":
            assert np.abs(out - ref).max() <= 1e-5 * ref.max()
    out = R.calculate_sampled_return_multi(0.9997, **base, n_step_samples=g["t_samples"], use_log_interpolation=True)
    assert np.array_equal(out, g["t_fast_exp_log"])


def test_get_return_estimate_all_distributions_and_modes_bit_exact(gold):
    g, meta = gold
    small = _base(g, "g_")
    for c in meta["g_cases"]:
        out = R.get_return_estimate(c["distribution"], c["mode"], 0.99, **small, n_step=6, max_samples=5,
                                    use_log_interpolation=c["log"], seed=7)
        assert np.array_equal(out, g["g_out_" + c["tag"]]), (c["tag"], np.abs(out - g["g_out_" + c["tag"]]).max())
    with pytest.raises(ValueError):
        R.get_return_estimate("banana", "standard", 0.99, **small)
    with pytest.raises(ValueError):
        R.get_return_estimate("uniform", "banana", 0.99, **small)


def test_config_size_against_oracle_and_device_tensors():
    """N = A = 256 with K = V = 108 geometric heads (SURVEY.md §8d TVF size: 56.6 MB of traffic)."""
    from ppo_amd.tvf import get_value_head_horizons
    rng = np.random.default_rng(1)
    N, A = 256, 256
    hz = get_value_head_horizons(108, 30000)
    K = V = len(hz)
    rewards = rng.normal(size=(N, A)).astype(np.float32)
    dones = rng.random((N, A)) < 0.01
    vs = rng.normal(size=(N + 1, A, V)).astype(np.float32)
    vs[:, :, 0] = 0
    np.random.seed(3)
    out = R.get_return_estimate("exponential", "advanced", 0.999, rewards, dones, hz, hz, vs, n_step=20, max_samples=8)
    np.random.seed(3)
    ref = T.get_return_estimate("exponential", "advanced", 0.999, rewards, dones, hz, hz, vs, n_step=20, max_samples=8)
    assert np.array_equal(out, ref)
    # device tensors in -> device tensor out
    np.random.seed(3)
    out_t = R.get_return_estimate("exponential", "advanced", 0.999, torch.from_numpy(rewards).cuda(),
                                  torch.from_numpy(dones).cuda(), hz, hz, torch.from_numpy(vs).cuda(), n_step=20,
                                  max_samples=8)
    assert out_t.is_cuda and torch.equal(out_t.cpu(), torch.from_numpy(ref))


def test_edges_zero_horizon_all_done_and_errors():
    rng = np.random.default_rng(2)
    N, A = 12, 5
    hz_v = np.asarray([0, 1, 2, 4, 8, 16])
    hz_r = np.asarray([0, 1, 3, 16])
    rewards = rng.normal(size=(N, A)).astype(np.float32)
    vs = rng.normal(size=(N + 1, A, len(hz_v))).astype(np.float32)
    for dones in (np.ones((N, A), bool), np.zeros((N, A), bool)):
        samp = rng.integers(1, N + 1, size=(len(hz_r), 3))
        out = R.calculate_sampled_return_multi(0.97, rewards, dones, hz_r, hz_v, vs, samp)
        assert np.array_equal(out, T.sampled_returns(0.97, rewards, dones, hz_r, hz_v, vs, samp))
        assert (out[:, :, 0] == 0).all()  # h = 0 has value 0 by definition (:580-583)
    with pytest.raises(IndexError):  # a target beyond the largest value horizon (the reference raises too)
        R.calculate_sampled_return_multi(0.97, rewards, np.zeros((N, A), bool), np.asarray([40]), hz_v, vs, np.asarray([[1]]))
    with pytest.raises(ValueError):
        R.calculate_sampled_return_multi(0.97, rewards, np.zeros((N, A + 1), bool), hz_r, hz_v, vs, np.ones((4, 1), int))


@pytest.mark.parametrize("N,A,K,C,what", [
    (40, 7, 12, 3, "remainder loop of the lane-resident records (C not a multiple of 4)"),
    (40, 7, 12, 12, "more than 8 samples per head: records by scalar loads"),
    (64, 5, 140, 8, "more than 128 heads: 16 result registers per lane, two register sets of records"),
    (300, 3, 20, 4, "N not a multiple of the chunk: partial last chunk, idle lanes"),
    (700, 2, 64, 2, "a column that does not fit in LDS: the prefix + gather fallback kernels"),
    (256, 3, 48, 8, "more than 128 distinct n-steps with lane-resident records: nd * T * 8 passes 65535"),
])
def test_kernel_paths_bit_exact_vs_oracle(N, A, K, C, what):
    """Every code path of csrc/tvf_returns.hip against the NumPy oracle, bit for bit."""
    rng = np.random.default_rng(N * 1000 + K)
    hz = np.unique(np.concatenate([[0, 1, 2, 3], np.geomspace(4, 5000, K - 4).astype(np.int64)]))
    hz = np.concatenate([hz, hz[-1] + 1 + np.arange(K - len(hz))]).astype(np.int64)  # K distinct, increasing
    assert len(hz) == K and (np.diff(hz) > 0).all()
    rewards = rng.normal(size=(N, A)).astype(np.float32)
    dones = rng.random((N, A)) < 0.03
    vs = rng.normal(size=(N + 1, A, K)).astype(np.float32)
    wide = "distinct n-steps" in what
    samples = rng.integers(1, (N if wide else min(N, 90)) + 1, size=(K, C))
    if wide:  # the (S, D) table's byte offset of the last n-step no longer fits in 16 bits (round 3's packing wrapped)
        nd = len(np.unique(np.minimum(samples, hz[:, None])[hz > 0]))
        assert nd >= 129 and C <= 8, nd
    out = R.calculate_sampled_return_multi(0.99, rewards, dones, hz, hz, vs, samples)
    ref = T.sampled_returns(0.99, rewards, dones, hz, hz, vs, samples)
    assert np.array_equal(out, ref), (what, np.abs(out - ref).max())
