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


@pytest.fixture(autouse=True, params=[1, 0], ids=["fused-mlp", "op-by-op"])
def mlp_path(request, monkeypatch):
    """Every test of this file runs twice: MLP nets on the fused launches (csrc/mlp_fused.hip, the default) and on the
    op-by-op path - both against the same reference fixtures, same bars."""
    monkeypatch.setattr(models, "FUSE_MLP", request.param)
    return request.param


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
    assert model.policy_net.mlp_fused == bool(models.FUSE_MLP) and model.value_net.mlp_fused == bool(models.FUSE_MLP)
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
    # Initial weights and the random mask are compared where they are generated - on the host, tests/test_model_init.py:
    # torch's CPU generators (and LAPACK's QR behind the orthogonal heads) only reproduce across hosts that take the same
    # vector code path, and the GPU box's host is not the fixture's.  Here the reference's initial weights are loaded
    # and, if this host drew another sparsity mask, the reference's mask takes its place.
    val = model.value_net
    mask = MASK_GOLD[f"{tag}_mask"]
    for prefix, net in (("policy_net", model.policy_net), ("value_net", model.value_net)):
        assert net.tvf_features_mask.shape == mask.shape and net.tvf_features_mask.dtype == torch.uint8
        if tag == "window":
            assert np.array_equal(net.tvf_features_mask.cpu().numpy(), mask)  # deterministic
        net.tvf_features_mask.copy_(cuda(mask))
        net.load_state_dict({name: torch.from_numpy(MASK_GOLD[f"{tag}_init_{prefix}.{name}"]) for name in net.state_dict()})
    assert 0 < mask.sum() < mask.size
    assert np.array_equal(val.params["tvf_head.weight"].cpu().numpy(), MASK_GOLD[f"{tag}_w0"])
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


def test_fused_mlp_reads_rows_through_the_index_and_steps_like_the_op_by_op_path(mlp_path):
    """The fused path's extras, on the Humanoid-shaped variant: (1) the minibatch read out of the WHOLE batch through the
    permutation gives bit-identical gradients to the gathered minibatch; (2) the statistics' column sums written by the
    weight-gradient launch equal the sum of the per-sample rows; (3) one optimiser step from the per-workgroup sums of
    g^2 it leaves (ppo_adam_step_presummed_f32) lands on the op-by-op path's parameters."""
    if not mlp_path:
        pytest.skip("fused path only")
    tag = "humanoid"
    model, m, _ = build(tag)
    val = model.value_net
    x = cuda(GOLD[f"{tag}_x"])
    MB = x.shape[0]
    g = torch.Generator(device="cuda").manual_seed(3)
    big = torch.randn(4 * MB, x.shape[1], device="cuda", generator=g)
    idx = torch.randperm(4 * MB, device="cuda", generator=g)[:MB].int().contiguous()
    big[idx.long()] = x
    ret, tvf_ret = cuda(GOLD[f"{tag}_value_returns"]), cuda(GOLD[f"{tag}_value_tvf_returns"])
    ret_big = torch.zeros(4 * MB, ret.shape[1], device="cuda")
    tvf_big = torch.zeros(4 * MB, tvf_ret.shape[1], device="cuda")
    ret_big[idx.long()], tvf_big[idx.long()] = ret, tvf_ret
    w = cuda(GOLD[f"{tag}_tvf_weights"])
    kw = dict(tvf_weights=w, vf_coef=m["ppo_vf_coef"], tvf_coef=m["tvf_coef"])
    val.grad.zero_()
    val.value_minibatch(x, returns=ret, tvf_returns=tvf_ret, **kw)
    g_gathered = val.grad.clone()
    val.grad.zero_()
    sums = torch.full((4,), 7.0, device="cuda")
    stats = val.value_minibatch(big, returns=ret_big, tvf_returns=tvf_big, index=idx, stat_sums=sums, **kw)
    assert torch.equal(val.grad, g_gathered)
    check_grads(val, tag, "value", m)
    assert torch.allclose(sums, stats.sum(0), rtol=1e-5, atol=1e-6)  # overwritten, not added to
    val.value_minibatch(big, returns=ret_big, tvf_returns=tvf_big, index=idx, stat_sums=sums, stat_accumulate=True, **kw)
    assert torch.allclose(sums, 2 * stats.sum(0), rtol=1e-5, atol=1e-6)
    # the optimiser step from the launch's own sums of g^2 against the separate sum-of-squares launch; a small
    # max_grad_norm so that the clip factor (the only consumer of the norm) matters
    before = val.flat.clone()
    norm_a, norm_b = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    assert val._presummed > 0
    val.adam_step(lr=1e-3, max_grad_norm=0.05, grad_norm_out=norm_a)
    after_a = val.flat.clone()
    val.flat.copy_(before)
    val.exp_avg.zero_(), val.exp_avg_sq.zero_()
    val._adam_step = 0
    assert val._presummed == 0
    val.adam_step(lr=1e-3, max_grad_norm=0.05, grad_norm_out=norm_b)
    assert float(norm_b) > 0.05 and abs(float(norm_a) - float(norm_b)) <= 2e-6 * float(norm_b)
    assert float((after_a - val.flat).abs().max()) <= 2e-6 * float(val.flat.abs().max())
    assert not torch.equal(after_a, before)
