"""GPU: ppo_amd.returns (HIP, through the C ABI) against the oracle and the
reference's golden vectors.

Parity bar:
  * columns regime: BIT-EXACT for every terminals dtype (same operation order
    and precision as the reference's loop, rl/returns.py:22-28);
  * tiles regime (float64 affine-map composition): max|x - ref| <= 1e-5 * max|ref|,
    the reference's own criterion (tests/test_tvf.py:46); for bool terminals
    (the production dtype, float64 carry in the reference) additionally
    <= 2 ulp of f32 per element.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import returns as O  # noqa: E402  (checker only)
from ppo_amd import _lib, returns as R  # noqa: E402

COLUMNS, TILES = _lib.PPO_SCAN_COLUMNS, _lib.PPO_SCAN_TILES


def _term(term, kind):
    return {"bool": term, "f32": term.astype(np.float32), "none": None}[kind]


def _close(x, ref, rel=1e-5):
    scale = max(np.abs(ref).max(), 1e-30)
    return np.abs(x.astype(np.float64) - ref.astype(np.float64)).max() <= rel * scale


def _ulp_diff(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib).max() if a.size else 0


@pytest.fixture(scope="module")
def gold(golden_dir):
    g = np.load(os.path.join(golden_dir, "returns_golden.npz"))
    meta = json.load(open(os.path.join(golden_dir, "returns_golden.json")))
    return g, meta


def test_library_is_loaded_and_gpu_present():
    lib = _lib.load()
    assert lib.ppo_version() >= 1
    assert torch.cuda.is_available()


def test_known_answer(gold):
    g, _ = gold
    for regime in (COLUMNS, TILES):
        out = R.gae(g["kat_r"], g["kat_v"], g["kat_vf"], g["kat_d"], 0.5, 1.0, regime=regime)
        assert np.abs(out - g["kat_expected"]).max() < 1e-4
    out = R.gae(g["kat_r"], g["kat_v"], g["kat_vf"], g["kat_d"], 0.5, 1.0, regime=COLUMNS)
    assert np.array_equal(out, g["kat_gae"])


@pytest.mark.parametrize("kind", ["bool", "f32", "none"])
def test_columns_regime_bit_exact_vs_reference_golden(gold, kind):
    g, meta = gold
    for c in meta["cases"]:
        k = c["key"]
        r, v, vf, term = g[k + "_r"], g[k + "_v"], g[k + "_vf"], g[k + "_term"]
        for j, (gamma, lamb) in enumerate(meta["gamma_lambda"]):
            t = _term(term, kind)
            assert np.array_equal(R.gae(r, v, vf, t, gamma, lamb, regime=COLUMNS), g[f"{k}_{kind}_{j}_gae"]), (k, j)
            assert np.array_equal(R.td_lambda(r, v, vf, t, gamma, lamb, regime=COLUMNS), g[f"{k}_{kind}_{j}_tdl"]), (k, j)


@pytest.mark.parametrize("kind", ["bool", "f32", "none"])
def test_tiles_regime_vs_reference_golden(gold, kind):
    g, meta = gold
    for c in meta["cases"]:
        k = c["key"]
        r, v, vf, term = g[k + "_r"], g[k + "_v"], g[k + "_vf"], g[k + "_term"]
        for j, (gamma, lamb) in enumerate(meta["gamma_lambda"]):
            t = _term(term, kind)
            a = R.gae(r, v, vf, t, gamma, lamb, regime=TILES)
            b = R.td_lambda(r, v, vf, t, gamma, lamb, regime=TILES)
            assert _close(a, g[f"{k}_{kind}_{j}_gae"]), (k, j)
            assert _close(b, g[f"{k}_{kind}_{j}_tdl"]), (k, j)
            if kind == "bool":
                assert _ulp_diff(a, g[f"{k}_{kind}_{j}_gae"]) <= 2, (k, j)


@pytest.mark.parametrize("regime", [COLUMNS, TILES])
def test_fused_pair_matches_separate_calls(regime):
    rng = np.random.default_rng(3)
    N, A = 64, 40
    r = rng.normal(size=(N, A)).astype(np.float32)
    v = rng.normal(size=(N + 1, A)).astype(np.float32)
    d = rng.random((N, A)) < 0.05
    adv, ret = R.gae_and_returns(r, v[:N], v[N], d, 0.999, 0.95, 0.8, regime=regime)
    assert np.array_equal(adv, R.gae(r, v[:N], v[N], d, 0.999, 0.95, regime=regime))
    assert np.array_equal(ret, R.td_lambda(r, v[:N], v[N], d, 0.999, 0.8, regime=regime))
    oa, orr = O.gae_and_returns(r, v[:N], v[N], d, 0.999, 0.95, 0.8)
    if regime == COLUMNS:
        assert np.array_equal(adv, oa) and np.array_equal(ret, orr)
    else:
        assert _close(adv, oa) and _close(ret, orr)


@pytest.mark.parametrize("shape", [(1, 1), (1, 5), (7, 1), (3, 63), (300, 17), (256, 256), (128, 1024),
                                   (257, 4100), (2, 9000), (1000, 24), (5, 40000)])
@pytest.mark.parametrize("kind", ["bool", "f32", "none"])
def test_ragged_shapes_both_regimes(shape, kind):
    N, A = shape
    rng = np.random.default_rng(N * 100003 + A)
    r = rng.normal(size=(N, A)).astype(np.float32)
    v = rng.normal(size=(N, A)).astype(np.float32)
    vf = rng.normal(size=(A,)).astype(np.float32)
    t = _term(rng.random((N, A)) < 0.03, kind)
    oa, orr = O.gae_and_returns(r, v, vf, t, 0.999, 0.95, 0.9)
    a, b = R.gae_and_returns(r, v, vf, t, 0.999, 0.95, 0.9, regime=COLUMNS)
    assert np.array_equal(a, oa) and np.array_equal(b, orr)
    a, b = R.gae_and_returns(r, v, vf, t, 0.999, 0.95, 0.9, regime=TILES)
    assert _close(a, oa) and _close(b, orr)
    a, b = R.gae_and_returns(r, v, vf, t, 0.999, 0.95, 0.9)  # auto
    assert _close(a, oa) and _close(b, orr)


def test_empty_inputs():
    for shape in [(0, 8), (8, 0), (0, 0)]:
        r = np.zeros(shape, np.float32)
        out = R.gae(r, r, np.zeros(shape[1], np.float32), None, 0.99, 0.95)
        assert out.shape == shape and out.dtype == np.float32


def test_all_done_and_no_done_edges():
    rng = np.random.default_rng(5)
    N, A = 50, 12
    r = rng.normal(size=(N, A)).astype(np.float32)
    v = rng.normal(size=(N, A)).astype(np.float32)
    vf = rng.normal(size=(A,)).astype(np.float32)
    for d in (np.ones((N, A), bool), np.zeros((N, A), bool)):
        for regime in (COLUMNS, TILES):
            a = R.gae(r, v, vf, d, 0.99, 0.95, regime=regime)
            assert _close(a, O.gae(r, v, vf, d, 0.99, 0.95))
    # every step terminal => advantage is the one-step TD error r - v
    a = R.gae(r, v, vf, np.ones((N, A), bool), 0.99, 0.95, regime=COLUMNS)
    assert np.array_equal(a, r - v)


def test_device_tensors_stay_on_device_and_inputs_untouched():
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(0)
    N, A = 32, 128
    r = torch.randn(N, A, generator=g).to(dev)
    value = torch.randn(N + 1, A, 1, generator=g).to(dev)  # Runner's [N+1, A, VH] buffer
    d = (torch.rand(N, A, generator=g) < 0.05).to(dev)
    r0, v0, d0 = r.clone(), value.clone(), d.clone()
    adv, ret = R.gae_and_returns(r, value[:N, :, 0], value[N, :, 0], d, 0.999, 0.95, 0.95)
    assert adv.is_cuda and ret.is_cuda and adv.dtype == torch.float32
    assert torch.equal(r, r0) and torch.equal(value, v0) and torch.equal(d, d0)
    oa, orr = O.gae_and_returns(r.cpu().numpy(), value[:N, :, 0].cpu().numpy(), value[N, :, 0].cpu().numpy(),
                                d.cpu().numpy(), 0.999, 0.95, 0.95)
    assert _close(adv.cpu().numpy(), oa) and _close(ret.cpu().numpy(), orr)


def test_errors_are_loud():
    r = np.zeros((4, 4), np.float32)
    with pytest.raises(ValueError):
        R.gae(r, r, np.zeros(3, np.float32), None, 0.9, 0.9)
    with pytest.raises(ValueError):
        R.gae(r, r, np.zeros(4, np.float32), np.zeros((4, 5), bool), 0.9, 0.9)
    with pytest.raises(_lib.PpoAmdError):
        R.gae(r, r, np.zeros(4, np.float32), None, 0.9, 0.9, regime=77)


@pytest.mark.parametrize("kind", ["bool", "f32"])
def test_bootstrapped_returns_bit_exact(gold, kind):
    g, meta = gold
    for c in meta["cases"]:
        k = c["key"]
        r, vf, term = g[k + "_r"], g[k + "_vf"], g[k + "_term"]
        for j, (gamma, _) in enumerate(meta["gamma_lambda"]):
            out = R.calculate_bootstrapped_returns(r, _term(term, kind), vf, gamma)
            assert np.array_equal(out, g[f"{k}_{kind}_{j}_boot"]), (k, j)
        if kind == "bool":
            out = R.calculate_bootstrapped_returns(r, term, vf, g[k + "_garr"])
            assert np.array_equal(out, g[k + "_boot_garr"]), k


def test_full_size_bandwidth_regime_sampled_columns():
    """BASELINE size N=256, A=2^20 (4.56 GB of traffic).  Columns are independent
    (rl/returns.py:22-28 never mixes envs), so the oracle on a sample of columns
    pins the full-size launch: sampled columns must be bit-exact, and a checksum
    over all columns of the tiles regime must agree with the columns regime."""
    dev = torch.device("cuda")
    N, A = 256, 1 << 20
    g = torch.Generator(device=dev).manual_seed(0)
    r = torch.randn(N, A, generator=g, device=dev)
    v = torch.randn(N + 1, A, generator=g, device=dev)
    d = torch.rand(N, A, generator=g, device=dev) < 0.01
    adv, ret = R.gae_and_returns(r, v[:N], v[N], d, 0.999, 0.95, 0.95)  # auto -> columns
    cols = torch.cat([torch.arange(0, 8), torch.randint(0, A, (500,), generator=torch.Generator().manual_seed(1)),
                      torch.arange(A - 8, A)]).to(dev)
    oa, orr = O.gae_and_returns(r[:, cols].cpu().numpy(), v[:N][:, cols].cpu().numpy(), v[N][cols].cpu().numpy(),
                                d[:, cols].cpu().numpy(), 0.999, 0.95, 0.95)
    assert np.array_equal(adv[:, cols].cpu().numpy(), oa)
    assert np.array_equal(ret[:, cols].cpu().numpy(), orr)
    # td_lambda = gae + value when both lambdas are equal (rl/returns.py:66-67)
    assert torch.equal(ret, adv + v[:N])
    adv_t, ret_t = R.gae_and_returns(r, v[:N], v[N], d, 0.999, 0.95, 0.95, regime=TILES)
    scale = adv.abs().max().item()
    assert (adv_t - adv).abs().max().item() <= 1e-5 * scale
    assert (ret_t - ret).abs().max().item() <= 1e-5 * ret.abs().max().item()
