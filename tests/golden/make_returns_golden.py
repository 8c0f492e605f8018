#!/usr/bin/env python3
"""Generate tests/golden/returns_golden.npz by running the REFERENCE's rl/returns.py.

Run in the build container only (needs /root/reference; it never travels):
    python tests/golden/make_returns_golden.py

rl/returns.py is pure NumPy, so it is loaded by file path with no shim.  The
fixture holds inputs and the reference's outputs (data only, no source):

  case_<i>_{r,v,vf,term}         inputs ([N,A] f32, [N,A] f32, [A] f32, [N,A] bool)
  case_<i>_<kind>_<j>_{gae,tdl}  outputs of gae / td_lambda for (gamma, lamb) #j
  case_<i>_<kind>_<j>_boot       calculate_bootstrapped_returns (kinds bool,f32)
  case_<i>_boot_garr             ... with an [N,A] gamma array
  kat_*                          the GAE known answer of rl/unit_tests.py:203-210
"""
import importlib.util
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/rl/returns.py"

GAMMA_LAMBDA = [(0.5, 1.0), (0.99, 0.95), (0.999, 0.95), (0.99997, 0.6), (1.0, 1.0), (0.999, 0.0)]
SHAPES = [(5, 1), (16, 8), (128, 16), (256, 64), (33, 7), (1, 3)]
PATTERNS = ["bernoulli", "none", "all", "first_last"]


def load_ref():
    spec = importlib.util.spec_from_file_location("ref_returns", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def make_term(rng, N, A, pattern):
    if pattern == "bernoulli":
        return rng.random((N, A)) < 0.05
    if pattern == "none":
        return np.zeros((N, A), bool)
    if pattern == "all":
        return np.ones((N, A), bool)
    t = np.zeros((N, A), bool)
    t[0, :] = True
    t[-1, ::2] = True
    return t


def main():
    ref = load_ref()
    rng = np.random.default_rng(20250131)
    out = {}
    meta = {"gamma_lambda": GAMMA_LAMBDA, "cases": []}
    ci = 0
    for (N, A) in SHAPES:
        for pattern in PATTERNS:
            if pattern != "bernoulli" and (N, A) not in [(16, 8), (33, 7)]:
                continue
            r = rng.normal(0, 1, (N, A)).astype(np.float32)
            v = rng.normal(0, 1, (N, A)).astype(np.float32)
            vf = rng.normal(0, 1, (A,)).astype(np.float32)
            term = make_term(rng, N, A, pattern)
            key = f"case_{ci}"
            out[f"{key}_r"], out[f"{key}_v"], out[f"{key}_vf"], out[f"{key}_term"] = r, v, vf, term
            for kind in ("bool", "f32", "none"):
                t = {"bool": term, "f32": term.astype(np.float32), "none": None}[kind]
                for j, (g, l) in enumerate(GAMMA_LAMBDA):
                    out[f"{key}_{kind}_{j}_gae"] = ref.gae(r, v, vf, t, g, l)
                    out[f"{key}_{kind}_{j}_tdl"] = ref.td_lambda(r, v, vf, t, g, l)
                    if kind != "none":
                        out[f"{key}_{kind}_{j}_boot"] = ref.calculate_bootstrapped_returns(r, t, vf, g)
            garr = rng.uniform(0.9, 1.0, (N, A)).astype(np.float32)
            out[f"{key}_garr"] = garr
            out[f"{key}_boot_garr"] = ref.calculate_bootstrapped_returns(r, term, vf, garr)
            meta["cases"].append({"key": key, "N": N, "A": A, "pattern": pattern})
            ci += 1

    # known answer, rl/unit_tests.py:203-210
    kr = np.asarray([1, 0, 2, 4, 6], dtype=np.float32)[:, None]
    kd = np.asarray([0, 0, 1, 0, 0], dtype=np.float32)[:, None]
    kv = np.asarray([0, 0.5, 0.5, 3, 4], dtype=np.float32)[:, None]
    kf = np.asarray(5, dtype=np.float32)
    out["kat_r"], out["kat_d"], out["kat_v"], out["kat_vf"] = kr, kd, kv, kf
    out["kat_gae"] = ref.gae(kr, kv, kf, kd, gamma=0.5, lamb=1.0)
    out["kat_expected"] = np.asarray([1.5, 0.5, 1.5, 5.25, 4.5])[:, None]
    for k, val in out.items():
        assert val.dtype != object, k
    np.savez_compressed(os.path.join(HERE, "returns_golden.npz"), **out)
    with open(os.path.join(HERE, "returns_golden.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", len(out), "arrays,", ci, "cases")


if __name__ == "__main__":
    main()
