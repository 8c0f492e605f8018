"""CPU: the oracle (oracle/returns_oracle.c) against the reference's own numbers.

Pins the oracle: the GAE known answer from the reference's rl/unit_tests.py:203-210
and the golden vectors produced by importing the reference's rl/returns.py
(tests/golden/make_returns_golden.py).  Bit-exact: the oracle reproduces NumPy's
promotion rules, so there is no tolerance here.
"""
import json
import os

import numpy as np
import pytest

from oracle import returns as O


@pytest.fixture(scope="module")
def gold(golden_dir):
    g = np.load(os.path.join(golden_dir, "returns_golden.npz"))
    meta = json.load(open(os.path.join(golden_dir, "returns_golden.json")))
    return g, meta


def _term(term, kind):
    return {"bool": term, "f32": term.astype(np.float32), "none": None}[kind]


def test_gae_known_answer(gold):
    g, _ = gold
    out = O.gae(g["kat_r"], g["kat_v"], g["kat_vf"], g["kat_d"], 0.5, 1.0)
    # reference tolerance for this KAT is 1e-4 abs (rl/unit_tests.py:11); we are exact
    assert np.array_equal(out, g["kat_gae"])
    assert np.abs(out - g["kat_expected"]).max() < 1e-4
    # bool terminals take the float64 path and must give the same answer here
    out_b = O.gae(g["kat_r"], g["kat_v"], g["kat_vf"], g["kat_d"].astype(bool), 0.5, 1.0)
    assert np.abs(out_b - g["kat_expected"]).max() < 1e-4


@pytest.mark.parametrize("kind", ["bool", "f32", "none"])
def test_gae_td_lambda_bit_exact(gold, kind):
    g, meta = gold
    for c in meta["cases"]:
        k = c["key"]
        r, v, vf, term = g[k + "_r"], g[k + "_v"], g[k + "_vf"], g[k + "_term"]
        for j, (gamma, lamb) in enumerate(meta["gamma_lambda"]):
            assert np.array_equal(O.gae(r, v, vf, _term(term, kind), gamma, lamb), g[f"{k}_{kind}_{j}_gae"]), (k, j)
            assert np.array_equal(O.td_lambda(r, v, vf, _term(term, kind), gamma, lamb), g[f"{k}_{kind}_{j}_tdl"]), (k, j)


def test_fused_pair_equals_two_calls(gold):
    g, meta = gold
    c = meta["cases"][2]["key"]
    r, v, vf, term = g[c + "_r"], g[c + "_v"], g[c + "_vf"], g[c + "_term"]
    adv, ret = O.gae_and_returns(r, v, vf, term, 0.999, 0.95, 0.6)
    assert np.array_equal(adv, O.gae(r, v, vf, term, 0.999, 0.95))
    assert np.array_equal(ret, O.td_lambda(r, v, vf, term, 0.999, 0.6))


@pytest.mark.parametrize("kind", ["bool", "f32"])
def test_bootstrapped_returns_bit_exact(gold, kind):
    g, meta = gold
    for c in meta["cases"]:
        k = c["key"]
        r, vf, term = g[k + "_r"], g[k + "_vf"], g[k + "_term"]
        for j, (gamma, _) in enumerate(meta["gamma_lambda"]):
            out = O.calculate_bootstrapped_returns(r, _term(term, kind), vf, gamma)
            assert np.array_equal(out, g[f"{k}_{kind}_{j}_boot"]), (k, j)
        if kind == "bool":
            out = O.calculate_bootstrapped_returns(r, term, vf, g[k + "_garr"])
            assert np.array_equal(out, g[k + "_boot_garr"]), k
