#!/usr/bin/env python3
"""Generate tests/golden/tvf_trim_golden.npz by running the REFERENCE's TVFRunnerModule.trim_horizons
(rl/tvf.py:91-208) on CPU.  Build container only (needs /root/reference; see ref_shim.py):

    python tests/golden/make_tvf_trim_golden.py

Cases: every trimming method (timelimit, est_term) x mode (interpolate, average, substitute, random), with and
without trim_clip, two horizon sets, env times spread around the time limit (some envs untrimmed, some with a few
steps left, some past the limit).  Data only: inputs (value estimates, times, episode-length buffer, horizons,
flags, np.random seed) and the three outputs."""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_shim import load_reference  # noqa: E402


def main():
    load_reference(["--tvf_enabled=True", "--device=cpu", "--env_reward_normalization=off", "--output_folder=/tmp/ref_golden_out"])
    from rl import config, tvf
    args = config.args
    out, meta = {}, {"cases": []}
    rng = np.random.default_rng(11)
    A, VH = 24, 1
    case = 0
    for n_heads, max_h, timeout in ((16, 1000, 1000), (48, 30000, 27000)):
        horizons = tvf.get_value_head_horizons(n_heads, max_h)
        K = len(horizons)
        out[f"horizons_{n_heads}"] = np.asarray(horizons)
        for method in ("timelimit", "est_term"):
            for mode in ("interpolate", "average", "substitute", "random"):
                for clip in (-1.0, 0.05):
                    args.env.timeout = timeout
                    type(args.tvf).trim_clip = clip
                    values = rng.normal(size=(A, K, VH)).astype(np.float32).cumsum(axis=1).astype(np.float32)
                    left = np.concatenate([rng.integers(0, 40, 8), rng.integers(40, timeout // 2, 8),
                                           [0, 1, 2, -3, -50, timeout - 1, timeout, timeout + 5][:8]])
                    time = (timeout - left).astype(np.int32)
                    buffer = [int(x) for x in rng.integers(timeout // 4, timeout, 30)]
                    log = types.SimpleNamespace(watch_stats=lambda *a, **k: None, watch_mean=lambda *a, **k: None)
                    runner = types.SimpleNamespace(tvf_horizons=horizons, K=K, episode_length_buffer=list(buffer), log=log)
                    mod = object.__new__(tvf.TVFRunnerModule)
                    mod.runner = runner
                    seed = 1000 + case
                    np.random.seed(seed)
                    trimmed, final, ttt = mod.trim_horizons(values, time, method=method, mode=mode)
                    tag = f"c{case}"
                    out[tag + "_values"], out[tag + "_time"], out[tag + "_buffer"] = values, time, np.asarray(buffer)
                    out[tag + "_trimmed"], out[tag + "_final"], out[tag + "_ttt"] = np.asarray(trimmed), np.asarray(final), np.asarray(ttt)
                    meta["cases"].append({"tag": tag, "n_heads": n_heads, "timeout": timeout, "method": method, "mode": mode,
                                          "trim_clip": clip, "seed": seed, "eta_percentile": args.tvf.eta_percentile,
                                          "eta_buffer": args.tvf.eta_buffer, "eta_minh": args.tvf.eta_minh,
                                          "trimmed_dtype": str(np.asarray(trimmed).dtype), "ttt_dtype": str(np.asarray(ttt).dtype)})
                    case += 1
    # method off: only h = 0 is zeroed
    values = rng.normal(size=(A, len(horizons), VH)).astype(np.float32)
    mod = object.__new__(tvf.TVFRunnerModule)
    mod.runner = types.SimpleNamespace(tvf_horizons=horizons, K=len(horizons), episode_length_buffer=[], log=None)
    trimmed, final, ttt = mod.trim_horizons(values, np.zeros(A, np.int32), method="off")
    out["off_values"], out["off_trimmed"] = values, trimmed
    meta["off"] = {"final": final, "ttt": ttt, "n_heads": n_heads}
    np.savez_compressed(os.path.join(HERE, "tvf_trim_golden.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "tvf_trim_golden.json"), "w"), indent=1)
    print("wrote", len(out), "arrays,", case, "cases;", sum(v.nbytes for v in out.values()) / 1e3, "KB raw")


if __name__ == "__main__":
    main()
