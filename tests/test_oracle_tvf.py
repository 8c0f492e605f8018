"""CPU: the TVF oracle (oracle/returns_truncated.py) and the product's host-side TVF helpers
(ppo_amd/tvf.py, ppo_amd.returns_truncated.interpolation_plan) against the reference's outputs in
tests/golden/tvf_golden.npz and the reference's own known answers (tests/test_tvf.py:10-129)."""
import json
import os

import numpy as np
import pytest

from oracle import returns_truncated as T


@pytest.fixture(scope="module")
def gold(golden_dir):
    return (np.load(os.path.join(golden_dir, "tvf_golden.npz")),
            json.load(open(os.path.join(golden_dir, "tvf_golden.json"))))


def _base(g, prefix):
    return {k: g[prefix + k] for k in ("rewards", "dones", "required_horizons", "value_sample_horizons", "value_samples")}


def test_oracle_equals_reference_fast_path_bitwise(gold):
    g, _ = gold
    base = _base(g, "t_")
    K = len(base["required_horizons"])
    for name, samp in (("n1", np.full((K, 1), 1)), ("n8", np.full((K, 1), 8)), ("n128", np.full((K, 1), 128)),
                       ("exp", g["t_samples"])):
        out = T.sampled_returns(0.9997, **base, n_step_samples=samp)
        assert np.array_equal(out, g["t_fast_" + name]), name
        # the reference's own check: fast vs its slow reference estimator, 1e-5 of the max (tests/test_tvf.py:46)
        ref = g["t_slow_" + name]
        assert np.abs(out - ref).max() <= 1e-5 * ref.max() or name == "n128"
    out = T.sampled_returns(0.9997, **base, n_step_samples=g["t_samples"], use_log_interpolation=True)
    assert np.array_equal(out, g["t_fast_exp_log"])


def test_oracle_get_return_estimate_all_modes_bitwise(gold):
    g, meta = gold
    small = _base(g, "g_")
    for c in meta["g_cases"]:
        out = T.get_return_estimate(c["distribution"], c["mode"], 0.99, **small, n_step=6, max_samples=5,
                                    use_log_interpolation=c["log"], seed=7)
        assert np.array_equal(out, g["g_out_" + c["tag"]]), c["tag"]
    with pytest.raises(ValueError):
        T.get_return_estimate("banana", "standard", 0.99, **small)
    with pytest.raises(ValueError):
        T.get_return_estimate("uniform", "banana", 0.99, **small)


def test_horizon_helpers_oracle_and_product(gold):
    g, _ = gold
    from ppo_amd import tvf as P
    for mod in (T, P):
        out = mod.horizon_interpolate(g["hi_kat_horizons"], g["hi_kat_values"], g["hi_kat_targets"])
        assert np.abs(out - g["hi_kat_expected"]).max() < 1e-6          # the reference's known answer
        assert np.array_equal(out, g["hi_kat_out"])
        out = mod.horizon_interpolate(g["hi_rand_horizons"], g["hi_rand_values"], g["hi_rand_targets"])
        assert np.array_equal(out, g["hi_rand_out"])
        for nh, mh in ((8, 1000), (128, 30000)):
            for sp in ("geometric", "linear"):
                h, w = mod.get_value_head_horizons(nh, mh, sp, include_weight=True)
                assert np.array_equal(h, g[f"vh_{nh}_{mh}_{sp}_h"]) and np.array_equal(w, g[f"vh_{nh}_{mh}_{sp}_w"])
                assert np.array_equal(mod.get_value_head_horizons(nh, mh, sp), h)
        with pytest.raises(ValueError):
            mod.get_value_head_horizons(8, 100, "banana")


def test_rediscounted_value_estimate_matches_reference(gold):
    """rl/tvf.py:388-433: equal gammas return the longest horizon; otherwise the per-interval reward mass is
    re-weighted by the (clipped) discount ratio.  Product (host-side, R8) against the reference's outputs."""
    g, _ = gold
    from ppo_amd import tvf as P
    for tag, (g_old, g_new) in {"same": (0.999, 0.999), "down": (0.9999, 0.99), "up": (0.99, 0.9999)}.items():
        got = P.get_rediscounted_value_estimate(g["rd_values"], g_old, g_new, g["rd_horizons"])
        assert got.dtype == g[f"rd_{tag}"].dtype and np.array_equal(got, g[f"rd_{tag}"]), tag
    assert np.array_equal(g["rd_same"], g["rd_values"][:, -1])
    assert (g["rd_down"] < g["rd_same"]).all() and (g["rd_up"] > g["rd_same"]).all()


def test_product_interpolation_plan_matches_oracle():
    from ppo_amd.returns_truncated import interpolation_plan as plan
    hz = np.asarray([0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 64])
    for target in range(-3, 65):
        o = T.interpolation_plan(hz, target)
        mode, i0, i1, w0, w1 = plan(hz, target)
        if o[0] == "zero":
            assert mode == 0
        elif o[0] == "exact":
            assert mode == 1 and i0 == o[1]
        else:
            assert mode == 2 and (i0, i1) == (o[1], o[2]) and w0 == float(1 - o[3]) and w1 == float(o[3])
    with pytest.raises(IndexError):
        plan(hz, 65)


def test_trim_horizons_matches_reference_bitwise(golden_dir):
    """ppo_amd.tvf.trim_horizons against the reference's TVFRunnerModule.trim_horizons (rl/tvf.py:91-208) run on CPU
    (tests/golden/make_tvf_trim_golden.py): every method x mode, with and without trim_clip, two horizon sets —
    trimmed estimates, the advantage value (mean over valid horizons) and the time till termination, bit for bit."""
    import json
    from ppo_amd import tvf
    g = np.load(os.path.join(golden_dir, "tvf_trim_golden.npz"))
    meta = json.load(open(os.path.join(golden_dir, "tvf_trim_golden.json")))
    assert len(meta["cases"]) == 32
    for c in meta["cases"]:
        t = c["tag"]
        np.random.seed(c["seed"])  # mode "random" draws from np.random in the reference's order
        trimmed, final, ttt = tvf.trim_horizons(
            g[f"horizons_{c['n_heads']}"], g[t + "_values"], g[t + "_time"], c["timeout"], method=c["method"],
            mode=c["mode"], trim_clip=c["trim_clip"], episode_lengths=g[t + "_buffer"].tolist(),
            eta_percentile=c["eta_percentile"], eta_buffer=c["eta_buffer"], eta_minh=c["eta_minh"])
        what = (c["method"], c["mode"], c["trim_clip"])
        assert trimmed.shape == g[t + "_trimmed"].shape and str(trimmed.dtype) == c["trimmed_dtype"], what
        assert np.array_equal(trimmed, g[t + "_trimmed"]), what
        assert final.dtype == np.float32 and np.array_equal(final, g[t + "_final"]), what
        assert str(np.asarray(ttt).dtype) == c["ttt_dtype"] and np.array_equal(ttt, g[t + "_ttt"]), what
        assert not np.array_equal(trimmed[..., 0], g[t + "_values"][..., 0]), "the case trims nothing"
    trimmed, final, ttt = tvf.trim_horizons(g[f"horizons_{meta['off']['n_heads']}"], g["off_values"],
                                            np.zeros(len(g["off_values"]), np.int32), 1000, method="off")
    assert np.array_equal(trimmed, g["off_trimmed"]) and final == 0 and ttt is None
    with pytest.raises(ValueError):
        tvf.trim_horizons(g["horizons_16"], g["c0_values"], g["c0_time"], 1000, method="nope")
    with pytest.raises(ValueError):
        tvf.trim_horizons(g["horizons_16"], g["c0_values"], g["c0_time"], 1000, mode="nope")
