"""Observation normalisation (`--observation_normalization`, rl/models.py:661-694).

CPU: the oracle restatement against the reference's own outputs (tests/golden/obsnorm_golden.npz).
GPU: the HIP kernels through the C ABI against the oracle and the golden vectors — the transform bit-exact
given equal constants, the running statistics to float32 rounding of one batch mean (the reference reduces a
batch in float32, the kernel in float64) — then the model and Runner paths that use them.
"""
import os

import numpy as np
import pytest

from oracle import obs_norm as ON  # checker

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "obsnorm_golden.npz"))
CASES = {"mlp": (11,), "img": (2, 12, 12)}
STAT_RTOL = 2e-6  # float32 batch mean / var (reference) vs float64 reduction (oracle, kernel)


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_reproduces_the_reference_normaliser(tag):
    n = ON.ObsNormalizer(CASES[tag], norm_eps=float(G[f"{tag}_norm_eps"]))
    assert n.rms.count == float(G[f"{tag}_count_init"])
    for i in range(3):
        x = G[f"{tag}_x{i}"]
        n.update(x)
        assert n.rms.count == float(G[f"{tag}_count{i}"])
        np.testing.assert_allclose(n.rms.mean, G[f"{tag}_mean{i}"], rtol=STAT_RTOL, atol=1e-7)
        np.testing.assert_allclose(n.rms.var, G[f"{tag}_var{i}"], rtol=STAT_RTOL, atol=1e-9)
        np.testing.assert_allclose(n.mu, G[f"{tag}_mu{i}"], rtol=STAT_RTOL, atol=1e-7)
        np.testing.assert_allclose(n.std, G[f"{tag}_std{i}"], rtol=STAT_RTOL, atol=1e-7)
        # the transform itself is exact once the constants are the reference's
        n.mu, n.std = G[f"{tag}_mu{i}"], G[f"{tag}_std{i}"]
        assert np.array_equal(n.apply(x), G[f"{tag}_y{i}"])
    assert np.array_equal(n.apply(G[f"{tag}_x3"]), G[f"{tag}_y3"])
    y = G[f"{tag}_y3"]
    assert y.min() >= -5 and y.max() <= 5 and y.dtype == np.float32


# ------------------------------------------------------------------------------------------------ GPU
torch = pytest.importorskip("torch")


def _exact_update(rms, x):
    """RunningMeanStd fed with the float64 batch moments (no float32 rounding in between): what the kernels
    compute, so they are held to it far tighter than to the float32-reducing reference."""
    xp = ON.prep(x).astype(np.float64)
    rms.update_from_moments(xp.mean(axis=0), xp.var(axis=0), xp.shape[0])


def _normalizer(dims, eps):
    from ppo_amd import models
    return models.ObsNormalizer(dims, "cuda", norm_eps=eps)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(CASES))
def test_kernels_match_the_golden_vectors(tag):
    n = _normalizer(CASES[tag], float(G[f"{tag}_norm_eps"]))
    o = ON.ObsNormalizer(CASES[tag], norm_eps=float(G[f"{tag}_norm_eps"]))
    for i in range(3):
        x = torch.from_numpy(G[f"{tag}_x{i}"]).cuda()
        n.update(x)
        _exact_update(o.rms, G[f"{tag}_x{i}"])
        assert n.count == float(G[f"{tag}_count{i}"])
        shape = CASES[tag]
        mean, var = n.mean.cpu().numpy().reshape(shape), n.var.cpu().numpy().reshape(shape)
        np.testing.assert_allclose(mean, G[f"{tag}_mean{i}"], rtol=STAT_RTOL, atol=1e-7)
        np.testing.assert_allclose(var, G[f"{tag}_var{i}"], rtol=STAT_RTOL, atol=1e-9)
        # against the update fed with float64 batch moments the statistics agree far tighter
        np.testing.assert_allclose(mean, o.rms.mean, rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(var, o.rms.var, rtol=1e-9, atol=1e-14)
        np.testing.assert_allclose(n.mu.cpu().numpy().reshape(shape), G[f"{tag}_mu{i}"], rtol=STAT_RTOL, atol=1e-7)
        np.testing.assert_allclose(n.std.cpu().numpy().reshape(shape), G[f"{tag}_std{i}"], rtol=STAT_RTOL, atol=1e-7)
        # transform: bit-exact given the reference's constants
        n.mu.copy_(torch.from_numpy(G[f"{tag}_mu{i}"]).reshape(-1))
        n.std.copy_(torch.from_numpy(G[f"{tag}_std{i}"]).reshape(-1))
        y = n.apply(x, torch.empty(x.shape, dtype=torch.float32, device="cuda"))
        assert np.array_equal(y.cpu().numpy(), G[f"{tag}_y{i}"])
    x3 = torch.from_numpy(G[f"{tag}_x3"]).cuda()
    assert np.array_equal(n.apply(x3, torch.empty(x3.shape, dtype=torch.float32, device="cuda")).cpu().numpy(), G[f"{tag}_y3"])


@pytest.mark.gpu
def test_kernels_at_the_atari_shape_and_ragged_feature_counts():
    rng = np.random.default_rng(0)
    for dims, u8 in (((4, 84, 84), True), ((3, 64, 64), True), ((17,), False), ((1, 5, 7), True), ((376,), False)):
        n, o = _normalizer(dims, 1e-5), ON.ObsNormalizer(dims, norm_eps=1e-5)
        for b in (256, 3):
            x = (rng.integers(0, 256, size=(b, *dims), dtype=np.uint8) if u8
                 else (rng.normal(size=(b, *dims)) * 7 + 3).astype(np.float32))
            xt = torch.from_numpy(x).cuda()
            n.update(xt)
            _exact_update(o.rms, x)
            np.testing.assert_allclose(n.mean.cpu().numpy().reshape(dims), o.rms.mean, rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose(n.var.cpu().numpy().reshape(dims), o.rms.var, rtol=1e-9, atol=1e-14)
            o.mu, o.std = n.mu.cpu().numpy().reshape(dims), n.std.cpu().numpy().reshape(dims)
            y = n.apply(xt, torch.empty(xt.shape, dtype=torch.float32, device="cuda"))
            assert np.array_equal(y.cpu().numpy(), o.apply(x))
    # frozen statistics do not move (rl/models.py:681)
    from ppo_amd import models
    f = models.ObsNormalizer((17,), "cuda", frozen=True)
    f.update(torch.randn(8, 17, device="cuda"))
    assert f.count == 1e-4 and float(f.mean.abs().sum()) == 0.0


@pytest.mark.gpu
def test_mlp_model_forward_matches_the_reference_with_normalisation():
    from ppo_amd import models
    torch.manual_seed(5)  # the seed of make_obsnorm_golden.py: same initial weights (tests/test_model_init.py)
    m = models.TVFModel("mlp", input_dims=(11,), actions=3, device="cuda", architecture="single", hidden_units=64,
                        encoder_activation_fn="tanh", observation_normalization=True,
                        head_scale=float(G["mlp_head_scale"]), head_bias=bool(G["mlp_head_bias"]))
    for i in range(3):
        y = m.perform_normalization(G[f"mlp_x{i}"], update_normalization=True)
        # feature 0 has mean -5 and std 0.01: one float32 ulp of mu (5e-7) is 5e-5 after the division
        np.testing.assert_allclose(y.cpu().numpy(), G[f"mlp_y{i}"], rtol=1e-5, atol=1e-4)
    rms = m.obs_rms
    np.testing.assert_allclose(rms.mean, G["mlp_mean2"], rtol=STAT_RTOL, atol=1e-7)
    assert rms.count == float(G["mlp_count2"])
    out = m.forward(G["mlp_x3"], output="policy")
    np.testing.assert_allclose(out["log_policy"].cpu().numpy(), G["mlp_fwd_log_policy"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(out["value"].cpu().numpy(), G["mlp_fwd_value"], rtol=1e-4, atol=5e-5)
    # update_normalization=True inside forward moves the statistics (rl/models.py:783-784)
    c = m.obs_norm.count
    m.forward(G["mlp_x3"], output="policy", update_normalization=True)
    assert m.obs_norm.count == c + 9


@pytest.mark.gpu
def test_impala_net_with_normalisation_equals_the_net_fed_normalised_floats():
    """The normalised path is the plain network on clamp((x/255 - mu) / (std + eps)): forward rows and every
    gradient of a PPO minibatch are bit-identical to a twin net given that float tensor directly."""
    from ppo_amd import models
    torch.manual_seed(0)
    dims, nA, B = (4, 84, 84), 6, 32
    a = models.DualHeadNet("impala", dims, nA, hidden_units=256, head_scale=0.1, head_bias=True, device="cuda")
    b = models.DualHeadNet("impala", dims, nA, hidden_units=256, head_scale=0.1, head_bias=True, device="cuda")
    b.load_state_dict(a.state_dict())
    norm = models.ObsNormalizer(dims, "cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    for _ in range(3):
        norm.update(torch.randint(0, 256, (64, *dims), dtype=torch.uint8, device="cuda", generator=g))
    a.obs_norm = norm
    x = torch.randint(0, 256, (B, *dims), dtype=torch.uint8, device="cuda", generator=g)
    xn = norm.apply(x, torch.empty(x.shape, dtype=torch.float32, device="cuda"))
    assert float(xn.std()) > 0.5  # really normalised (uniform bytes -> unit variance), not x/255
    for _ in range(2):  # second pass replays the recorded inference plan
        ha = a.forward(x)["_heads"].clone()
        hb = b.forward(xn)["_heads"].clone()
        assert torch.equal(ha, hb)
    actions = torch.randint(0, nA, (B,), dtype=torch.int32, device="cuda", generator=g)
    logp = torch.log_softmax(torch.randn(B, nA, device="cuda", generator=g), dim=1)
    pac = logp.gather(1, actions.long()[:, None])[:, 0].contiguous()
    adv, ret = torch.randn(B, device="cuda", generator=g), torch.randn(B, device="cuda", generator=g)
    a.ppo_minibatch(x, actions, pac, logp, adv, ret)
    b.ppo_minibatch(xn, actions, pac, logp, adv, ret)
    torch.cuda.synchronize()
    assert torch.equal(a.grad, b.grad) and float(a.grad.abs().sum()) > 0


@pytest.mark.gpu
def test_runner_with_normalisation_trains_and_checkpoints(tmp_path):
    from ppo_amd import envs, logger, models, rollout
    from ppo_amd.config import args
    args.setup(["--agents=16", "--n_steps=32", "--model_architecture=single", "--model_encoder=mlp",
                "--model_hidden_units=64", "--env_type=classic", "--env_name=CartPole", "--seed=2", "--device=cuda",
                "--policy_opt_mini_batch_size=128", "--policy_opt_epochs=2", "--workers=2", "--gamma=0.99",
                "--disable_logging=True", "--observation_normalization=True"])
    torch.manual_seed(2)
    np.random.seed(2)
    model = models.TVFModel("mlp", input_dims=(4,), actions=2, device="cuda", architecture="single", hidden_units=64,
                            head_scale=0.1, head_bias=True, observation_normalization=True,
                            norm_eps=args.observation_normalization_epsilon)
    r = rollout.Runner(model, logger.Logger(quiet=True))
    r.vec_env = envs.create_envs_classic()
    try:
        r.reset()
        for _ in range(3):
            r.generate_rollout()
            r.calculate_returns()
            r.train()
        # one update per env step of the rollout, from all envs (rl/rollout.py:735-741); none for the final state
        assert abs(model.obs_norm.count - (1e-4 + 3 * 32 * 16)) < 1e-9
        # the statistics are those of the observations the rollouts saw
        seen = r.all_obs[:32].reshape(-1, 4).double()
        assert torch.isfinite(r.net.flat).all() and float(model.obs_norm.var.min()) > 0
        assert float((model.obs_norm.mean - seen.mean(0)).abs().max()) < 0.5
        path = r.save_checkpoint(str(tmp_path / "cp.pt"), r.step)
        mean, var, count = model.obs_norm.mean.clone(), model.obs_norm.var.clone(), model.obs_norm.count
        out_before = model.forward(r.all_obs[0], output="policy")["log_policy"].clone()
        model.obs_norm.update(torch.randn(64, 4, device="cuda") * 100)  # disturb, then restore
        assert not torch.equal(model.obs_norm.mean, mean)
        r.load_checkpoint(path)
        assert torch.equal(model.obs_norm.mean, mean) and torch.equal(model.obs_norm.var, var)
        assert model.obs_norm.count == count
        assert torch.equal(model.forward(r.all_obs[0], output="policy")["log_policy"], out_before)
    finally:
        r.vec_env.close()


@pytest.mark.gpu
def test_pipelined_rollout_with_normalisation_equals_the_generic_path():
    """With array-stepping envs the rollout keeps its per-group streams and host / GPU overlap under observation
    normalisation too: statistics from every env's step-t observation first, then the groups' forwards.  Same bytes as
    the one-group, one-sync-per-step path the reference's order was first built on."""
    from ppo_amd import logger, models, rollout
    from ppo_amd.config import args
    from ppo_amd.vec_env import SplitVecEnv, SyntheticVecEnv

    class GymOnly:  # hides step_arrays: the Runner takes the generic path
        def __init__(self, env):
            self.env, self.num_envs = env, env.num_envs

        def reset(self):
            return self.env.reset()

        def step(self, a):
            return self.env.step(a)

    args.setup(["--agents=32", "--n_steps=8", "--model_architecture=single", "--model_encoder=impala", "--env_type=synthetic",
                "--env_embed_time=False", "--seed=3", "--device=cuda", "--disable_logging=True",
                "--observation_normalization=True"])
    outs = []
    for mode in ("generic", "pipelined"):
        torch.manual_seed(3)
        model = models.TVFModel("impala", input_dims=(4, 84, 84), actions=6, device="cuda", architecture="single",
                                hidden_units=256, head_scale=0.1, head_bias=True, observation_normalization=True,
                                norm_eps=args.observation_normalization_epsilon)
        r = rollout.Runner(model, logger.Logger(quiet=True))
        if mode == "generic":
            r.vec_env = GymOnly(SyntheticVecEnv(32, seed=3, p_done=0.05, threads=2))
        else:
            r.vec_env = SplitVecEnv([SyntheticVecEnv(16, seed=3, p_done=0.05, env_offset=16 * i, threads=2) for i in range(2)])
        r.reset()
        r.generate_rollout()
        r.generate_rollout()
        torch.cuda.synchronize()
        assert abs(model.obs_norm.count - (1e-4 + 2 * 8 * 32)) < 1e-9
        outs.append([x.cpu().clone() for x in (r.all_obs, r.actions, r.log_policy, r.value, r.ext_rewards, r.terminals,
                                               model.obs_norm.mean, model.obs_norm.var, model.obs_norm.mu, model.obs_norm.std)]
                    + [torch.from_numpy(r.all_time.copy())])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert outs[0][5].any() and len(torch.unique(outs[0][1])) > 1
