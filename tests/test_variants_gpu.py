"""GPU: the network variants beyond IMPALA/single/discrete against the reference's own outputs
(tests/golden/variants_golden.npz, recorded by make_variants_golden.py from the reference on CPU):
MLP encoder, tanh / relu encoder activation, dual architecture routing, TVF heads, gaussian policy,
and one minibatch of each training phase — policy (rl/rollout.py:1610-1771), value (:1513-1567 with the
TVF loss of rl/tvf.py:32-77) and distillation (:1331-1449) — losses and every parameter gradient.

Tolerances: these nets are small dense layers (no ReLU/max-pool kinks at 64 units matter at this size for
tanh; the relu variant can flip a kink), f32 both sides: forward 2e-6, gradients 2e-5 of the largest entry."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from ppo_amd import models  # noqa: E402

HERE = os.path.dirname(__file__)
GOLD = np.load(os.path.join(HERE, "golden", "variants_golden.npz"))
META = json.load(open(os.path.join(HERE, "golden", "variants_golden.json")))


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def build(tag):
    m = META[tag]
    tvf = f"{tag}_tvf_horizons" in GOLD
    torch.manual_seed(7)
    model = models.TVFModel(
        "mlp", input_dims=tuple(m["input_dims"]), actions=m["n_actions"], device="cuda", architecture="dual",
        hidden_units=m["hidden"], encoder_activation_fn=m["activation"], head_scale=m["head_scale"],
        head_bias=m["head_bias"], tvf_fixed_head_horizons=list(GOLD[f"{tag}_tvf_horizons"]) if tvf else None,
        tvf_fixed_head_weights=list(GOLD[f"{tag}_tvf_weights"]) if tvf else None)
    if f"{tag}_log_std" in GOLD:
        model.policy_net.params["log_std"].copy_(cuda(GOLD[f"{tag}_log_std"]))
    return model, m, tvf


def close(a, want, tol, what):
    a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    scale = max(float(np.abs(want).max()), 1e-6)
    err = float(np.abs(a.reshape(want.shape) - want).max())
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def check_grads(net, tag, phase, m, tol=2e-5):
    none = set(m.get("grad_none", {}).get(f"{tag}_{phase}", []))
    seen = 0
    for name in net.params:
        key = f"{tag}_{phase}_grad_{name}"
        if key in GOLD:
            close(net.grads[name], GOLD[key], tol, f"{phase} grad {name}")
            seen += 1
        else:
            assert name in none, name
            # the reference leaves these gradients as None; here they are exact zeros
            assert float(net.grads[name].abs().max()) == 0.0, f"{phase}: {name} should get no gradient"
    assert seen >= 6


@pytest.mark.parametrize("tag", ["mlp_gauss_tvf", "mlp_disc", "humanoid"])
def test_state_dict_keys_and_forward_routing(tag):
    model, m, tvf = build(tag)
    assert list(model.state_dict().keys()) == m["state_dict_keys"]
    x = cuda(GOLD[f"{tag}_x"])
    for mode in ("default", "full", "policy", "value"):
        r = model.forward(x, output=mode)
        assert sorted(r.keys()) == m["forward_keys"][mode], mode
        for k, v in r.items():
            close(v, GOLD[f"{tag}_fwd_{mode}_{k}"], 2e-6, f"{mode}/{k}")
    if tvf:
        assert r["tvf_value"].shape == (x.shape[0], len(GOLD[f"{tag}_tvf_horizons"]), 1)
        sub = model.forward(x, output="value", required_tvf_heads=[0, 3])["tvf_value"]
        close(sub, GOLD[f"{tag}_fwd_value_tvf_value"][:, [0, 3]], 2e-6, "required_tvf_heads")


@pytest.mark.parametrize("tag", ["mlp_gauss_tvf", "humanoid"])
def test_gaussian_policy_value_and_distil_minibatches(tag):
    """`humanoid` is BASELINE configs[4] at its real size: 377 float observations, 256 tanh units, 17 gaussian
    actions, 128 TVF heads out to horizon 30 000."""
    model, m, _ = build(tag)
    x = cuda(GOLD[f"{tag}_x"])
    pol, val = model.policy_net, model.value_net
    pol.grad.zero_()
    stats = pol.gaussian_minibatch(x, cuda(GOLD[f"{tag}_policy_actions"]), cuda(GOLD[f"{tag}_policy_log_pac"]),
                                   cuda(GOLD[f"{tag}_policy_advantages"]), None, eps_clip=m["ppo_epsilon"])
    res = GOLD[f"{tag}_policy_result"]
    s = stats.cpu().numpy().astype(np.float64)
    assert abs(-s[:, 6].mean() - res[0]) < 2e-6 and abs(s[:, 3].mean() - res[3]) < 1e-6
    check_grads(pol, tag, "policy", m)

    val.grad.zero_()
    stats = val.value_minibatch(x, returns=cuda(GOLD[f"{tag}_value_returns"]),
                                tvf_returns=cuda(GOLD[f"{tag}_value_tvf_returns"]),
                                tvf_weights=cuda(GOLD[f"{tag}_tvf_weights"]), vf_coef=m["ppo_vf_coef"],
                                tvf_coef=m["tvf_coef"])
    s = stats.cpu().numpy().astype(np.float64)
    res = GOLD[f"{tag}_value_result"]
    assert abs(s[:, 2].mean() - res[0]) < 2e-6 * max(1, abs(res[0]))
    assert abs(s[:, 2].std(ddof=1) - res[1]) < 1e-5 * max(1, abs(res[1]))
    check_grads(val, tag, "value", m)

    pol.grad.zero_()
    stats = pol.distil_minibatch(x, cuda(GOLD[f"{tag}_distil_distil_targets"]), cuda(GOLD[f"{tag}_distil_old_raw_policy"]),
                                 beta=m["distil_beta"], use_tvf=True, weights=cuda(GOLD[f"{tag}_tvf_weights"]),
                                 gaussian=True)
    s = stats.cpu().numpy().astype(np.float64)
    res = GOLD[f"{tag}_distil_result"]
    assert abs(s[:, 2].mean() - res[0]) < 2e-6 * max(1, abs(res[0]))
    check_grads(pol, tag, "distil", m)


def test_h_weighted_tvf_value_loss():
    """--tvf_head_weighting=h_weighted (rl/tvf.py:55-62) is a per-head weight vector on the same kernel."""
    import types

    from ppo_amd import tvf
    from ppo_amd.config import args
    tag = "mlp_gauss_tvf"
    model, m, _ = build(tag)
    args.setup(["--tvf_enabled=True", "--tvf_max_horizon=1000", "--tvf_head_weighting=h_weighted"])
    stub = types.SimpleNamespace(head_weighting="h_weighted",
                                 runner=types.SimpleNamespace(tvf_weights=GOLD[f"{tag}_tvf_weights"],
                                                              tvf_horizons=GOLD[f"{tag}_tvf_horizons"]))
    w = tvf.TVFRunnerModule.value_loss_weights(stub)
    stub.head_weighting = "off"
    assert np.array_equal(tvf.TVFRunnerModule.value_loss_weights(stub), GOLD[f"{tag}_tvf_weights"].astype(np.float32))
    args.setup([])
    assert w.dtype == np.float32 and w[0] > w[-1]  # short horizons weigh more
    val = model.value_net
    val.grad.zero_()
    stats = val.value_minibatch(cuda(GOLD[f"{tag}_x"]), returns=cuda(GOLD[f"{tag}_value_returns"]),
                                tvf_returns=cuda(GOLD[f"{tag}_value_tvf_returns"]), tvf_weights=cuda(w),
                                vf_coef=m["ppo_vf_coef"], tvf_coef=m["tvf_coef"])
    s = stats.cpu().numpy().astype(np.float64)
    res = GOLD[f"{tag}_valuehw_result"]
    assert abs(res[0] - GOLD[f"{tag}_value_result"][0]) > 1e-3  # the weighting really changes the loss
    assert abs(s[:, 2].mean() - res[0]) < 2e-6 * max(1, abs(res[0]))
    assert abs(s[:, 2].std(ddof=1) - res[1]) < 1e-5 * max(1, abs(res[1]))
    check_grads(val, tag, "valuehw", m)


def test_discrete_policy_value_and_distil_minibatches():
    tag = "mlp_disc"
    model, m, _ = build(tag)
    x = cuda(GOLD[f"{tag}_x"])
    pol, val = model.policy_net, model.value_net
    pol.grad.zero_()
    stats = pol.ppo_minibatch(x, cuda(GOLD[f"{tag}_policy_actions"]).int(), cuda(GOLD[f"{tag}_policy_log_pac"]),
                              cuda(GOLD[f"{tag}_policy_log_policy"]), cuda(GOLD[f"{tag}_policy_advantages"]), None,
                              eps_clip=m["ppo_epsilon"], ent_coef=m["entropy_bonus"], vf_coef=0.0)
    s = stats.cpu().numpy().astype(np.float64)
    res = GOLD[f"{tag}_policy_result"]
    assert abs(-s[:, 6].mean() - res[0]) < 2e-6 and abs(s[:, 4].mean() - res[1]) < 2e-6
    assert abs(s[:, 5].mean() - res[2]) < 2e-6 and abs(s[:, 3].mean() - res[3]) < 1e-6
    check_grads(pol, tag, "policy", m)

    val.grad.zero_()
    stats = val.value_minibatch(x, returns=cuda(GOLD[f"{tag}_value_returns"]), vf_coef=m["ppo_vf_coef"])
    s = stats.cpu().numpy().astype(np.float64)
    assert abs(s[:, 2].mean() - GOLD[f"{tag}_value_result"][0]) < 2e-6
    check_grads(val, tag, "value", m)

    pol.grad.zero_()
    stats = pol.distil_minibatch(x, cuda(GOLD[f"{tag}_distil_distil_targets"]), cuda(GOLD[f"{tag}_distil_old_log_policy"]),
                                 beta=m["distil_beta"])
    s = stats.cpu().numpy().astype(np.float64)
    assert abs(s[:, 2].mean() - GOLD[f"{tag}_distil_result"][0]) < 2e-6
    check_grads(pol, tag, "distil", m)


def test_gaussian_sampling_statistics_and_log_prob():
    """ppo_gaussian_act_f32: a = mu + sigma n with n ~ N(0,1) (moments over 64k draws), log_pac equal to
    torch's Normal.log_prob of the drawn actions, deterministic mode returns mu, fixed `normal` is honoured."""
    from ppo_amd import _lib
    lib = _lib.load()
    B, nA, vh = 4096, 16, 1
    ldo = 2 * nA + vh
    g = torch.Generator(device="cuda").manual_seed(0)
    heads = torch.randn(B, ldo, device="cuda", generator=g)
    log_std = torch.linspace(-1.0, 0.5, nA, device="cuda")
    outs = [torch.empty(B, nA, device="cuda") for _ in range(3)]
    values = torch.empty(B, vh, device="cuda")

    def act(normal, seed, offset, det):
        rc = lib.ppo_gaussian_act_f32(heads.data_ptr(), B, ldo, nA, log_std.data_ptr(),
                                      None if normal is None else normal.data_ptr(), seed, offset, det,
                                      outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), values.data_ptr(), vh,
                                      _lib.current_stream())
        _lib.check(rc, "ppo_gaussian_act_f32")
        torch.cuda.synchronize()
        return [o.clone() for o in outs]

    a, lp, mu = act(None, 123, 0, 0)
    assert torch.equal(mu, heads[:, :nA]) and torch.equal(values, heads[:, nA:nA + vh])
    n = ((a - mu) / log_std.exp()).cpu().numpy().astype(np.float64)
    assert abs(n.mean()) < 0.02 and abs(n.std() - 1.0) < 0.02 and abs((n ** 3).mean()) < 0.05
    assert abs((n ** 4).mean() - 3.0) < 0.15 and np.abs(n).max() < 6.5
    want = torch.distributions.Normal(mu, log_std.exp()).log_prob(a)
    assert torch.allclose(lp, want, atol=2e-5, rtol=1e-5)
    a2, _, _ = act(None, 123, 0, 0)
    a3, _, _ = act(None, 123, B * nA, 0)
    assert torch.equal(a, a2) and not torch.equal(a, a3)  # counter-based: same (seed, offset) -> same draw
    det, _, _ = act(None, 123, 0, 1)
    assert torch.equal(det, mu)
    fixed = torch.randn(B, nA, device="cuda", generator=g)
    af, _, _ = act(fixed, 0, 0, 0)
    assert torch.allclose(af, mu + log_std.exp() * fixed, atol=1e-6)


def test_tanh_kernels_match_torch():
    from ppo_amd import _lib
    lib = _lib.load()
    x = torch.linspace(-9, 9, 100003, device="cuda")
    y, dx = torch.empty_like(x), torch.empty_like(x)
    dy = torch.randn_like(x)
    _lib.check(lib.ppo_tanh_forward_f32(x.data_ptr(), y.data_ptr(), x.numel(), _lib.current_stream()), "tanh")
    _lib.check(lib.ppo_tanh_backward_f32(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), x.numel(), _lib.current_stream()),
               "tanh bwd")
    torch.cuda.synchronize()
    assert float((y - torch.tanh(x)).abs().max()) <= 2e-7
    assert torch.allclose(dx, dy * (1 - y * y), atol=0, rtol=0)


MASK_GOLD = np.load(os.path.join(HERE, "golden", "tvf_mask_golden.npz"))
MASK_META = json.load(open(os.path.join(HERE, "golden", "tvf_mask_golden.json")))


@pytest.mark.parametrize("tag", ["sparsity", "window"])
def test_tvf_feature_masks_match_the_reference(tag):
    """--tvf_feature_sparsity / --tvf_feature_window (rl/models.py:386-427): a static mask over the TVF head's weights.
    From the same seed: the masked + rescaled initial head (sha256), the 0 / 1 mask, the forward; then one optimiser
    step with the reference's gradients - the step moves masked weights (the fixture counts them), the next forward must
    see them zeroed again, as the reference's mask_feature_weights does."""
    m = MASK_META[tag]
    torch.manual_seed(7)
    model = models.TVFModel(
        "mlp", input_dims=tuple(m["input_dims"]), actions=m["n_actions"], device="cuda", architecture="dual",
        hidden_units=m["hidden"], encoder_activation_fn="tanh", head_scale=m["head_scale"], head_bias=m["head_bias"],
        tvf_fixed_head_horizons=list(MASK_GOLD["horizons"]), tvf_fixed_head_weights=None,
        tvf_feature_sparsity=m["tvf_feature_sparsity"], tvf_feature_window=m["tvf_feature_window"])
    import hashlib
    for prefix, net in (("policy_net", model.policy_net), ("value_net", model.value_net)):
        for name, t in net.state_dict().items():
            a = np.ascontiguousarray(t.detach().cpu().numpy())
            assert hashlib.sha256(a.tobytes()).hexdigest() == m["params"][f"{prefix}.{name}"]["sha256"], (prefix, name)
    val = model.value_net
    mask = MASK_GOLD[f"{tag}_mask"]
    assert np.array_equal(val.tvf_features_mask.cpu().numpy(), mask) and 0 < mask.sum() < mask.size
    x = cuda(MASK_GOLD["x"])
    close(model.forward(x, output="value")["tvf_value"], MASK_GOLD[f"{tag}_fwd0_tvf_value"], 2e-6, "forward 0")
    # one optimiser step with the reference's gradients (plain Adam, no clipping: max_grad_norm 0)
    val.grad.zero_()
    for name in val.params:
        key = f"{tag}_grad_{name}"
        if key in MASK_GOLD:
            val.grads[name].copy_(cuda(MASK_GOLD[key]).reshape(val.grads[name].shape))
    assert m["masked_entries_nonzero_after_step"] > 0  # the step does move masked weights in the reference
    val.adam_step(lr=1e-2, eps=1e-5, max_grad_norm=0.0)
    w1 = val.params["tvf_head.weight"].cpu().numpy()
    assert (w1[mask == 0] == 0).all()
    close(w1, MASK_GOLD[f"{tag}_w1"], 2e-6, "head after step + mask")
    for name in val.params:
        close(val.params[name], MASK_GOLD[f"{tag}_after_{name}"] * (mask if name == "tvf_head.weight" else 1), 2e-6, name)
    close(model.forward(x, output="value")["tvf_value"], MASK_GOLD[f"{tag}_fwd1_tvf_value"], 5e-6, "forward 1")
    # a loaded state_dict is masked too
    sd = {k: v.clone() for k, v in val.state_dict().items()}
    sd["tvf_head.weight"] = torch.ones_like(sd["tvf_head.weight"])
    val.load_state_dict(sd)
    assert np.array_equal(val.params["tvf_head.weight"].cpu().numpy(), mask.astype(np.float32))
