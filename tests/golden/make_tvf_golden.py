#!/usr/bin/env python3
"""Generate tests/golden/tvf_golden.npz by running the REFERENCE's truncated-return code.

Build container only (needs /root/reference):  python tests/golden/make_tvf_golden.py

rl/returns_truncated.py is pure NumPy and is loaded by file path; rl/tvf.py needs the package import
and goes through ref_shim.  Captured (data only):
  t_*      the input recipe of the reference's own test (tests/test_tvf.py:14-24: N,A,K,V = 128,16,16,32,
           gamma 0.9997, rewards in {-1..2}, ~3 % dones, value_samples[:, :, 0] = 0) from a seeded generator,
           with the outputs of _calculate_sampled_return_multi_fast AND of the slow
           _calculate_sampled_return_multi_reference for n_step_list [1], [8], [128] and for an exponential
           per-horizon sample matrix (tests/test_tvf.py:109-118);
  g_*      get_return_estimate(seed=...) for every distribution x mode (small case), log interpolation on/off;
  hi_*     horizon_interpolate known answer (tests/test_tvf.py:121-129) and a random case;
  vh_*     get_value_head_horizons for (8, 1000) and (128, 30000) with weights, geometric and linear.
"""
import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def main():
    spec = importlib.util.spec_from_file_location("ref_rt", "/root/reference/rl/returns_truncated.py")
    rt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rt)
    from ref_shim import load_reference
    load_reference(["--device=cpu"])
    import rl.tvf as tvf

    out, meta = {}, {"g_cases": []}
    rng = np.random.default_rng(20250131)
    N, A, K, V = 128, 16, 16, 32
    base = {
        "gamma": 0.9997,
        "rewards": rng.integers(-1, 3, [N, A]).astype("float32"),
        "dones": rng.integers(0, 101, [N, A]) >= 98,
        "required_horizons": np.geomspace(1, 1024, num=K).astype("int32"),
        "value_sample_horizons": np.geomspace(1, 1024, num=V).astype("int32") - 1,
        "value_samples": rng.normal(0.1, 0.4, [N + 1, A, V]).astype("float32"),
    }
    base["value_samples"][:, :, 0] *= 0
    for k, v in base.items():
        out["t_" + k] = np.asarray(v)
    n_step, n_samples = 20, 8
    lamb = 1 - (1 / n_step)
    w = np.asarray([lamb ** x for x in range(128)], dtype=np.float32)
    samples = rng.choice(np.arange(1, len(w) + 1), [K, n_samples], replace=True, p=w / w.sum())
    out["t_samples"] = samples
    for name, kw in (("n1", {"n_step_list": [1]}), ("n8", {"n_step_list": [8]}), ("n128", {"n_step_list": [128]}),
                     ("exp", {"n_step_samples": samples})):
        out[f"t_fast_{name}"] = rt._calculate_sampled_return_multi_fast(**base, **kw)
        out[f"t_slow_{name}"] = rt._calculate_sampled_return_multi_reference(**base, **kw)
    out["t_fast_exp_log"] = rt._calculate_sampled_return_multi_fast(**base, n_step_samples=samples, use_log_interpolation=True)

    # get_return_estimate over distributions x modes, seeded
    n, a, k, v = 32, 4, 8, 12
    small = {
        "gamma": 0.99,
        "rewards": rng.normal(0, 1, [n, a]).astype("float32"),
        "dones": rng.random([n, a]) < 0.05,
        "required_horizons": np.asarray([0, 1, 2, 3, 5, 9, 17, 40], dtype="int32"),
        "value_sample_horizons": np.asarray([0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 64], dtype="int32"),
        "value_samples": rng.normal(0, 1, [n + 1, a, v]).astype("float32"),
    }
    small["value_samples"][:, :, 0] = 0
    for key, val in small.items():
        out["g_" + key] = np.asarray(val)
    for dist in ("fixed", "exponential", "uniform", "hyperbolic", "quadratic"):
        for mode in ("standard", "advanced", "clipped", "adaptive", "mcx", "full"):
            if dist == "fixed" and mode != "standard":
                continue
            for log in (False, True):
                if log and mode not in ("advanced", "full"):
                    continue
                tag = f"{dist}_{mode}_{int(log)}"
                out["g_out_" + tag] = rt.get_return_estimate(dist, mode, **small, n_step=6, max_samples=5,
                                                             use_log_interpolation=log, seed=7)
                meta["g_cases"].append({"tag": tag, "distribution": dist, "mode": mode, "log": log})

    # horizon_interpolate: the reference's known answer + a random case
    hz = np.asarray([0, 1, 2, 10, 100])
    vals = np.asarray([0, 5, 10, -1, 2])[None, :].repeat(11, axis=0)
    tg = np.asarray([-100, -1, 0, 1, 2, 3, 4, 99, 100, 101, 200])
    out["hi_kat_horizons"], out["hi_kat_values"], out["hi_kat_targets"] = hz, vals, tg
    out["hi_kat_out"] = tvf.horizon_interpolate(hz, vals, tg)
    out["hi_kat_expected"] = np.asarray([0, 0, 0, 5, 10, (7 / 8) * 10 + (1 / 8) * -1, (6 / 8) * 10 + (2 / 8) * -1,
                                         1.96666667, 2, 2, 2])
    hz2 = np.asarray([0, 1, 3, 7, 20, 50, 300])
    v2 = rng.normal(0, 1, [6, 5, len(hz2)]).astype("float32")
    t2 = rng.integers(-5, 400, [6, 5])
    out["hi_rand_horizons"], out["hi_rand_values"], out["hi_rand_targets"] = hz2, v2, t2
    out["hi_rand_out"] = tvf.horizon_interpolate(hz2, v2, t2)

    for nh, mh in ((8, 1000), (128, 30000)):
        for sp in ("geometric", "linear"):
            h, wts = tvf.get_value_head_horizons(nh, mh, sp, include_weight=True)
            out[f"vh_{nh}_{mh}_{sp}_h"], out[f"vh_{nh}_{mh}_{sp}_w"] = np.asarray(h), np.asarray(wts)
    # get_rediscounted_value_estimate (rl/tvf.py:388-433): cumulative per-horizon values, three gamma pairs
    rng_rd = np.random.default_rng(77)
    hz_rd = np.asarray(tvf.get_value_head_horizons(16, 3000, "geometric"))
    v_rd = np.cumsum(rng_rd.random((12, len(hz_rd))).astype(np.float32), axis=1)
    v_rd[:, 0] = 0
    out["rd_horizons"], out["rd_values"] = hz_rd, v_rd
    for tag, (g_old, g_new) in {"same": (0.999, 0.999), "down": (0.9999, 0.99), "up": (0.99, 0.9999)}.items():
        out[f"rd_{tag}"] = np.asarray(tvf.get_rediscounted_value_estimate(v_rd, g_old, g_new, hz_rd))
    np.savez_compressed(os.path.join(HERE, "tvf_golden.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "tvf_golden.json"), "w"), indent=1)
    print("wrote", len(out), "arrays", sum(x.nbytes for x in out.values()) / 1e6, "MB raw")
    print("fast vs slow max abs:", {n_: float(np.abs(out[f't_fast_{n_}'] - out[f't_slow_{n_}']).max()) for n_ in ("n1", "n8", "n128", "exp")})


if __name__ == "__main__":
    main()
