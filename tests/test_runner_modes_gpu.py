"""GPU: the Runner beyond IMPALA/single/discrete, end to end —
  * BASELINE.json configs[0]: CartPole, 8+ envs in the process-pool vector env, MLP policy, and PPO has
    to actually learn it (episode length grows);
  * dual architecture (DNA) on image observations: policy / value / distil phases each step their own
    optimiser, the distil phase starts from KL = 0;
  * configs[4]-shaped: gaussian MLP policy + TVF heads (dual): rollout buffers, GAE on the longest TVF
    horizon, TVF return targets equal to the oracle's on the same NumPy draws, all three phases run.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import returns as O  # noqa: E402 (checker)
from oracle import returns_truncated as OT  # noqa: E402 (checker)
from ppo_amd import envs, logger, models, rollout, tvf  # noqa: E402
from ppo_amd.config import args  # noqa: E402


def test_cartpole_mlp_learns_through_the_process_pool():
    args.setup(["--agents=16", "--n_steps=64", "--model_architecture=single", "--model_encoder=mlp",
                "--model_hidden_units=64", "--env_type=classic", "--env_name=CartPole", "--seed=2", "--device=cuda",
                "--policy_opt_mini_batch_size=256", "--policy_opt_epochs=4", "--policy_opt_lr=0.001", "--workers=4",
                "--gamma=0.99", "--disable_logging=True"])
    torch.manual_seed(2)
    np.random.seed(2)
    shape, nA = envs.get_env_spec()
    assert shape == (4,) and nA == 2
    model = models.TVFModel("mlp", input_dims=shape, actions=nA, device="cuda", architecture="single",
                            hidden_units=64, head_scale=0.1, head_bias=True)
    r = rollout.Runner(model, logger.Logger(quiet=True))
    r.vec_env = envs.create_envs_classic()
    try:
        from ppo_amd import wrappers
        assert wrappers.get_wrapper(r.vec_env, wrappers.VecNormalizeRewardWrapper) is not None  # rms by default
        r.reset()
        assert r.all_obs.dtype == torch.float32 and r.all_obs.shape == (65, 16, 4)
        lengths = []
        for it in range(30):
            r.generate_rollout()
            r.calculate_returns()
            r.train()
            # mean length of the episodes that ended in this rollout = steps / terminations
            done = int(r.terminals.sum())
            lengths.append(r.N * r.A / max(done, 1))
        stats = r.fetch_stats()
        assert np.isfinite(stats["loss_policy"]) and torch.isfinite(r.net.flat).all()
        norm = wrappers.get_wrapper(r.vec_env, wrappers.VecNormalizeRewardWrapper)
        assert r.reward_scale == pytest.approx(1.0 / float(norm.std)) and r.reward_scale < 1.0  # rl/rollout.py:1793
        early, late = np.mean(lengths[:3]), np.mean(lengths[-3:])
        assert early < 40, lengths  # random play: ~20 steps per episode
        assert late > 2.5 * early, lengths  # PPO learned to balance
    finally:
        r.vec_env.close()


def test_cartpole_learns_with_the_dual_architecture():
    """DNA end to end: policy from policy_net, advantages from value_net's estimates, value phase on value_net,
    distillation back into policy_net — PPO still has to learn CartPole with all of that in the loop."""
    args.setup(["--agents=16", "--n_steps=64", "--model_architecture=dual", "--model_encoder=mlp",
                "--model_hidden_units=64", "--env_type=classic", "--env_name=CartPole", "--seed=4", "--device=cuda",
                "--policy_opt_mini_batch_size=256", "--policy_opt_epochs=4", "--policy_opt_lr=0.001",
                "--value_opt_mini_batch_size=256", "--value_opt_epochs=2", "--value_opt_lr=0.001",
                "--distil_opt_mini_batch_size=256", "--distil_opt_epochs=1", "--workers=4", "--gamma=0.99",
                "--disable_logging=True"])
    torch.manual_seed(4)
    np.random.seed(4)
    model = models.TVFModel("mlp", input_dims=(4,), actions=2, device="cuda", architecture="dual", hidden_units=64,
                            head_scale=0.1, head_bias=True)
    r = rollout.Runner(model, logger.Logger(quiet=True))
    r.vec_env = envs.create_envs_classic()
    try:
        r.reset()
        lengths = []
        for _ in range(30):
            r.generate_rollout()
            r.calculate_returns()
            r.train()
            lengths.append(r.N * r.A / max(int(r.terminals.sum()), 1))
        stats = r.fetch_stats()
        assert np.isfinite(stats["loss_distil"]) and np.isfinite(stats["loss_value"])
        early, late = np.mean(lengths[:3]), np.mean(lengths[-3:])
        assert early < 40 and late > 2.5 * early, lengths
    finally:
        r.vec_env.close()


def test_dual_architecture_runs_policy_value_distil_phases():
    args.setup(["--agents=16", "--n_steps=16", "--model_architecture=dual", "--model_encoder=impala",
                "--env_type=synthetic", "--env_embed_time=False", "--seed=5", "--device=cuda",
                "--policy_opt_mini_batch_size=64", "--value_opt_mini_batch_size=64", "--distil_opt_mini_batch_size=64",
                "--disable_logging=True"])
    torch.manual_seed(5)
    np.random.seed(5)
    shape, nA = envs.get_env_spec()
    model = models.TVFModel("impala", input_dims=shape, actions=nA, device="cuda", architecture="dual",
                            hidden_units=256, head_scale=0.1, head_bias=True)
    assert model.name == "DNA-impala" and model.value_net is not model.policy_net
    r = rollout.Runner(model, logger.Logger(quiet=True))
    r.vec_env = envs.create_envs_classic()
    r.reset()
    r.generate_rollout()
    torch.cuda.synchronize()
    # the rollout's value estimates come from value_net, its policy from policy_net
    out = model.forward(r.all_obs[3])
    assert torch.allclose(out["value"], r.value[3], atol=1e-6) and torch.allclose(out["log_policy"], r.log_policy[3], atol=1e-6)
    vonly = model.value_net.forward(r.all_obs[3].contiguous())["value"]
    assert torch.allclose(vonly, r.value[3], atol=1e-6)
    assert not torch.allclose(model.policy_net.forward(r.all_obs[3].contiguous())["value"], r.value[3], atol=1e-4)
    r.calculate_returns()
    pol0, val0 = model.policy_net.flat.clone(), model.value_net.flat.clone()
    r.train()
    torch.cuda.synchronize()
    stats = r.fetch_stats()
    n_mb = 16 * 16 // 64
    assert model.policy_net._adam_step == 2 * n_mb  # policy epochs 2
    assert model.value_net._adam_step == 1 * n_mb  # value epochs 1
    assert r.distil_optimizer.state.step == 2 * n_mb  # distil epochs 2, its own Adam moments
    assert not torch.equal(pol0, model.policy_net.flat) and not torch.equal(val0, model.value_net.flat)
    for k in ("loss_pg", "loss_value", "loss_distil", "loss_distil_policy", "grad_value", "grad_distil"):
        assert np.isfinite(stats[k]), k
    assert stats["loss_v_ext"] > 0 and stats["loss_distil_value"] > 0
    # the distil phase starts at the policy it must stay close to: first minibatch KL is 0 to rounding
    first = float(r._phase_stats["distil"][0][0, 1].item())
    assert abs(first) < 1e-5 and stats["loss_distil_policy"] < 1e-3
    assert r.batch_counter == 1 and r.wants_distil_update("after_policy") and not r.wants_distil_update("before_policy")
    # second iteration
    r.generate_rollout()
    r.calculate_returns()
    r.train()
    torch.cuda.synchronize()
    assert torch.isfinite(model.policy_net.flat).all() and torch.isfinite(model.value_net.flat).all()


class FloatVecEnv:
    """Deterministic in-process vector env with flat float observations and continuous actions."""

    def __init__(self, A, dim, seed):
        self.num_envs, self.dim = A, dim
        self.rng = np.random.default_rng(seed)
        self.t = np.zeros(A, np.int64)

    def reset(self):
        self.t[:] = 0
        return self.rng.standard_normal((self.num_envs, self.dim)).astype(np.float32)

    def step(self, actions):
        assert actions.shape == (self.num_envs, 3) and actions.dtype == np.float32
        self.t += 1
        rew = (1.0 - 0.1 * np.square(actions).sum(1)).astype(np.float32)
        done = self.rng.random(self.num_envs) < 0.05
        infos = [{"time": int(t), "ep_length": int(t), "ep_score": float(t)} for t in self.t]
        self.t[done] = 0
        return self.rng.standard_normal((self.num_envs, self.dim)).astype(np.float32), rew, done, infos


def test_gaussian_tvf_dual_runner_matches_oracle_returns(tmp_path):
    args.setup(["--agents=8", "--n_steps=32", "--model_architecture=dual", "--model_encoder=mlp",
                "--model_hidden_units=64", "--env_type=mujoco", "--env_name=Fake", "--seed=9", "--device=cuda",
                "--tvf_enabled=True", "--tvf_value_heads=8", "--tvf_max_horizon=100", "--tvf_return_samples=4",
                "--policy_opt_mini_batch_size=64", "--value_opt_mini_batch_size=64", "--distil_opt_mini_batch_size=64",
                "--env_reward_normalization=off", "--disable_logging=True"])
    torch.manual_seed(9)
    np.random.seed(9)
    horizons, weights = tvf.get_value_head_horizons(args.tvf.value_heads, args.tvf.max_horizon, args.tvf.head_spacing,
                                                    include_weight=True)
    model = models.TVFModel("mlp", input_dims=(11,), actions=3, device="cuda", architecture="dual", hidden_units=64,
                            encoder_activation_fn="tanh", tvf_fixed_head_horizons=horizons,
                            tvf_fixed_head_weights=weights, head_scale=0.1, head_bias=True)
    assert model.name == "TVF-mlp"
    r = rollout.Runner(model, logger.Logger(quiet=True), action_dist="gaussian")
    r.vec_env = FloatVecEnv(8, 11, seed=4)
    r.reset()
    r.generate_rollout()
    torch.cuda.synchronize()
    N, A, K = r.N, r.A, len(horizons)
    assert r.actions.shape == (N, A, 3) and r.actions.dtype == torch.float32 and r.log_pac.shape == (N, A, 3)
    # log_pac is the log-density of the stored action under N(raw_policy, exp(log_std))
    std = torch.exp(model.policy_net.params["log_std"])
    want = torch.distributions.Normal(r.raw_policy, std).log_prob(r.actions)
    assert torch.allclose(r.log_pac, want, atol=2e-5)
    noise = (r.actions - r.raw_policy) / std
    assert abs(float(noise.mean())) < 0.15 and 0.8 < float(noise.std()) < 1.2
    # TVF estimates recorded from value_net
    tv = model.value_net.forward(r.all_obs[7].contiguous())["tvf_value"].clone()
    tv[:, 0] = 0  # the h = 0 head is zero by definition when recorded (rl/rollout.py:791)
    assert tv.shape == (A, K, 1) and torch.allclose(tv, r.tvf.tvf_value[7], atol=1e-6)
    assert float(r.tvf.tvf_value[:, :, 0].abs().max()) == 0.0
    # returns: GAE on the longest horizon, TVF targets = oracle on the same NumPy draws
    np.random.seed(123)
    r.calculate_returns()
    torch.cuda.synchronize()
    ext = r.tvf.tvf_value[:, :, -1, 0].cpu().numpy()
    rew, done = r.ext_rewards.cpu().numpy(), r.terminals.cpu().numpy()
    oa, orr = O.gae_and_returns(rew, ext[:N], ext[N], done, args.gamma, args.lambda_policy, args.lambda_value)
    assert np.abs(r.advantage.cpu().numpy() - oa).max() <= 1e-5 * np.abs(oa).max()
    assert np.abs(r.returns.view(N, A).cpu().numpy() - orr).max() <= 1e-5 * np.abs(orr).max()
    np.random.seed(123)
    want = OT.get_return_estimate(args.tvf.return_distribution, args.tvf.return_mode, args.tvf.gamma, rew, done,
                                  np.asarray(horizons), np.asarray(horizons), r.tvf.tvf_value[..., 0].cpu().numpy(),
                                  n_step=args.tvf_return_n_step, max_samples=args.tvf.return_samples)
    got = r.tvf.tvf_returns[..., 0].cpu().numpy()
    assert got.shape == want.shape == (N, A, K)
    assert np.abs(got - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-6)
    # all three phases
    pol0, val0 = model.policy_net.flat.clone(), model.value_net.flat.clone()
    log_std0 = model.policy_net.params["log_std"].clone()
    r.train()
    torch.cuda.synchronize()
    stats = r.fetch_stats()
    assert not torch.equal(pol0, model.policy_net.flat) and not torch.equal(val0, model.value_net.flat)
    assert not torch.equal(log_std0, model.policy_net.params["log_std"])  # the gaussian loss trains log_std
    assert stats["loss_tvf"] > 0 and np.isfinite(stats["loss_distil"]) and stats["entropy"] == 0.0
    assert torch.equal(model.value_net.params["log_std"], torch.zeros(3, device="cuda"))  # value net's is untouched
    # checkpoint round trip incl. value / distil optimiser states
    path = str(tmp_path / "cp.pt")
    r.save_checkpoint(path, 77)
    v = model.value_net.flat.clone()
    m = r.distil_optimizer.state.exp_avg.clone()
    model.value_net.flat.zero_()
    r.distil_optimizer.state.exp_avg.zero_()
    assert r.load_checkpoint(path) == 77
    assert torch.equal(model.value_net.flat, v) and torch.equal(r.distil_optimizer.state.exp_avg, m)
    out = r.detached_batch_forward(r.all_obs[:4].reshape(-1, 11), output="default", max_batch_size=16)
    assert out["tvf_value"].shape == (32, K, 1) and out["raw_policy"].shape == (32, 3)


def test_tvf_trimming_and_horizon_dropout_through_the_runner():
    """--tvf_trimming (rl/tvf.py:91-208 called per env step at rl/rollout.py:788-804, 884-892): the Runner records the
    untrimmed estimates and env times, trims the rollout step by step with the episode-length buffer growing as the
    rollout did, and GAE reads the value `--tvf_trim_advantages` names.  --tvf_horizon_dropout (rl/tvf.py:64-69):
    each (sample, head) TVF term kept with probability 1 - p and weighted 1 / (1 - p)."""
    args.setup(["--agents=8", "--n_steps=32", "--model_architecture=dual", "--model_encoder=mlp",
                "--model_hidden_units=64", "--env_type=mujoco", "--env_name=Fake", "--seed=9", "--device=cuda",
                "--tvf_enabled=True", "--tvf_value_heads=8", "--tvf_max_horizon=100", "--tvf_return_samples=4",
                "--tvf_trimming=est_term", "--tvf_trimming_mode=average", "--tvf_trim_advantages=average",
                "--tvf_eta_minh=4", "--tvf_eta_buffer=2", "--env_timeout=30", "--tvf_horizon_dropout=0.5",
                "--policy_opt_mini_batch_size=64", "--value_opt_mini_batch_size=64", "--distil_opt_mini_batch_size=64",
                "--env_reward_normalization=off", "--disable_logging=True"])
    torch.manual_seed(9)
    np.random.seed(9)
    horizons, weights = tvf.get_value_head_horizons(args.tvf.value_heads, args.tvf.max_horizon, args.tvf.head_spacing,
                                                    include_weight=True)
    model = models.TVFModel("mlp", input_dims=(11,), actions=3, device="cuda", architecture="dual", hidden_units=64,
                            encoder_activation_fn="tanh", tvf_fixed_head_horizons=horizons,
                            tvf_fixed_head_weights=weights, head_scale=0.1, head_bias=True)
    r = rollout.Runner(model, logger.Logger(quiet=True), action_dist="gaussian")
    r.vec_env = FloatVecEnv(8, 11, seed=4)
    r.reset()
    assert list(r.tvf.episode_length_buffer) == [1000]
    r.generate_rollout()
    r.generate_rollout()  # the second rollout starts with a warm episode-length buffer and non-zero env times
    torch.cuda.synchronize()
    N, A, K = r.N, r.A, len(horizons)
    assert r.tvf.tvf_untrimmed_value is not r.tvf.tvf_value
    untrimmed = r.tvf.tvf_untrimmed_value.cpu().numpy()
    assert r.all_time.shape == (N + 1, A) and r.all_time.max() > 10 and np.array_equal(r.all_time[N], r.time)
    # replay: the reference's per-step call, buffer as it stood at each step
    import collections
    finished_first = sum(len(x) for x in r._finished_lengths)
    assert finished_first > 0
    buf = collections.deque(list(r.tvf.episode_length_buffer)[:len(r.tvf.episode_length_buffer) - finished_first], maxlen=1000)
    trimmed_any = False
    for t in range(N + 1):
        want, final, _ = tvf.trim_horizons(horizons, untrimmed[t], r.all_time[t], 30, method="est_term", mode="average",
                                           trim_clip=-1, episode_lengths=buf, eta_percentile=args.tvf.eta_percentile,
                                           eta_buffer=2, eta_minh=4)
        assert np.array_equal(r.tvf.tvf_value[t].cpu().numpy(), want), t
        assert np.array_equal(r.tvf.tvf_final_value[t].cpu().numpy(), final), t
        trimmed_any |= not np.array_equal(want, untrimmed[t])
        if t < N:
            buf.extend(r._finished_lengths[t])
    assert trimmed_any, "no state came close enough to the time limit to be trimmed"
    # advantages from the mean over valid horizons
    r.calculate_returns()
    ext = r.tvf.tvf_final_value.cpu().numpy()
    oa, _ = O.gae_and_returns(r.ext_rewards.cpu().numpy(), ext[:N], ext[N], r.terminals.cpu().numpy(), args.gamma,
                              args.lambda_policy, args.lambda_value)
    assert np.abs(r.advantage.cpu().numpy() - oa).max() <= 1e-5 * np.abs(oa).max()
    # horizon dropout in the value phase: about half of the (sample, head) terms survive, each weighted 2x
    net = model.value_net
    B = 64
    obs = r.all_obs[:8].reshape(B, 11).contiguous()
    tvf_ret = r.tvf.tvf_returns[:8, :, :, 0].reshape(B, K).contiguous()
    w = torch.ones(K, device="cuda")
    net.value_minibatch(obs, tvf_returns=tvf_ret, tvf_weights=w, tvf_coef=1.0)
    full = net.last_dheads(B)[:, net.col_tvf:].clone()
    net.value_minibatch(obs, tvf_returns=tvf_ret, tvf_weights=w, tvf_coef=1.0, tvf_keep_prob=0.5, dropout_seed=5, dropout_offset=0)
    d1 = net.last_dheads(B)[:, net.col_tvf:].clone()
    net.value_minibatch(obs, tvf_returns=tvf_ret, tvf_weights=w, tvf_coef=1.0, tvf_keep_prob=0.5, dropout_seed=5, dropout_offset=B * K)
    d2 = net.last_dheads(B)[:, net.col_tvf:].clone()
    live = full != 0
    kept = (d1 != 0) & live
    assert 0.35 < float(kept.sum()) / float(live.sum()) < 0.65
    assert torch.allclose(d1[kept], 2 * full[kept], rtol=1e-6) and not torch.equal(d1, d2)
    r.train()
    torch.cuda.synchronize()
    stats = r.fetch_stats()
    assert np.isfinite(stats["loss_tvf"]) and stats["loss_tvf"] > 0 and r.tvf._dropout_calls > 0


def _fake_atari_env(seed, cfg):
    """Atari wrapper stack (ppo_amd.atari.make, the reference's rl/atari.py:119-230 order) over the scripted simulator."""
    import os
    import sys
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    if golden not in sys.path:
        sys.path.insert(0, golden)
    import fake_envs
    from ppo_amd import atari
    # cfg: the parsed flags travel with the constructor — env workers are spawned, their own `args` is unparsed
    return atari.make("FakeGame", seed=seed, args=cfg, base_env=fake_envs.FakeAtari())


def test_atari_wrapper_stack_through_the_process_pool_and_runner():
    """north_star: "vectorised envs (Atari/Procgen/MuJoCo via rl.hybridVecEnv)".  ALE is not installed, so the simulator
    is the scripted stand-in; everything above it is the product path: the Atari wrapper stack (no-op starts, frame
    skip with max, time limit, grey-scale 84x84, action marks, 4-frame stack + time channel, channels first, null
    action) in worker processes, the shared pinned observation block, rms reward normalisation, the IMPALA net at
    5 x 84 x 84."""
    import functools
    args.setup(["--agents=8", "--n_steps=32", "--model_architecture=single", "--model_encoder=impala", "--env_type=atari",
                "--env_name=FakeGame", "--env_timeout=40", "--env_noop_duration=4", "--seed=5", "--device=cuda",
                "--policy_opt_mini_batch_size=64", "--policy_opt_epochs=1", "--workers=4", "--disable_logging=True"])
    assert (args.env.frame_skip, args.env.frame_stack, args.env.color_mode) == (4, 4, "bw")
    shape, nA = envs.get_env_spec()
    assert shape == (5, 84, 84) and nA == 6
    torch.manual_seed(5)
    np.random.seed(5)
    model = models.TVFModel("impala", input_dims=shape, actions=nA, device="cuda", architecture="single",
                            hidden_units=256, head_scale=0.1, head_bias=True)
    r = rollout.Runner(model, logger.Logger(quiet=True))
    r.vec_env = envs.create_envs_classic(env_fns=[functools.partial(_fake_atari_env, 100 + 997 * i, args) for i in range(8)])
    try:
        r.reset()
        assert r.obs.shape == (8, 5, 84, 84) and r.obs.dtype == np.uint8
        for _ in range(2):
            r.generate_rollout()
            r.calculate_returns()
            r.train()
        torch.cuda.synchronize()
        obs = r.all_obs.cpu().numpy()
        assert obs.shape == (33, 8, 5, 84, 84) and obs[..., :4, :, :].max() > 0
        # the time channel: uint8(255 * steps / timeout) of the state, constant over the frame, 0 right after a reset
        tc = obs[:, :, 4]
        assert (tc == tc[:, :, :1, :1]).all() and tc.max() > 0
        assert np.array_equal(np.unique(tc[1:][r.terminals.cpu().numpy()]), [0])
        # episodes end by the 40-step time limit (or the simulator), rewards are normalised and finite
        assert int(r.terminals.sum()) >= 2 and r.all_time.max() <= 40
        assert torch.isfinite(r.ext_rewards).all() and torch.isfinite(r.net.flat).all() and r.net._adam_step == 8
    finally:
        r.vec_env.close()


def test_runner_logs_value_quality_every_fourth_batch():
    """rl/rollout.py:1252-1285: moments of advantages / returns / values every batch; feature statistics and
    explained variance when batch_counter % 4 == 3 (skipped under --disable_ev); the numbers are those of the
    runner's own buffers."""
    from ppo_amd import value_quality as vq
    args.setup(["--agents=16", "--n_steps=8", "--model_architecture=single", "--model_encoder=impala",
                "--env_type=synthetic", "--env_embed_time=False", "--seed=5", "--device=cuda",
                "--policy_opt_mini_batch_size=64", "--policy_opt_epochs=1"])
    torch.manual_seed(5)
    np.random.seed(5)
    shape, nA = envs.get_env_spec()
    model = models.TVFModel("impala", input_dims=shape, actions=nA, device="cuda", architecture="single",
                            hidden_units=256, head_scale=0.1, head_bias=True)
    log = logger.Logger(quiet=True)
    r = rollout.Runner(model, log)
    r.vec_env = envs.create_envs_classic()
    r.reset()
    for it in range(4):
        r.generate_rollout()
        r.calculate_returns()
        assert ("ev_ext" in log) == (it == 3), it
        assert log["adv_ext_mean"] == pytest.approx(float(r.advantage.double().mean()), abs=1e-9)
        assert log["*return_ext_std"] == pytest.approx(float(r.returns.double().std(unbiased=False)), rel=1e-9)
        r.train()
    N = r.N
    values, rewards = r.value[:N, :, 0].cpu().numpy(), r.ext_rewards.cpu().numpy()
    targets = O.calculate_bootstrapped_returns(rewards, r.terminals.cpu().numpy(), r.value[N, :, 0].cpu().numpy(), args.gamma)
    want = vq.explained_variance(values.ravel(), targets.ravel())
    assert log["ev_ext"] == pytest.approx(want, abs=1e-5) and log["ev_average"] == log["ev_ext"]
    assert log["z_target_var"] == pytest.approx(float(np.var(targets.astype(np.float64))), rel=1e-4)
    for key in ("*policy_features_sparsity", "*policy_raw_features_std", "reward_scale", "entropy_bonus", "*gamma",
                "value_ext_mean", "*ext_value_estimates_std"):
        assert key in log, key
    assert 0.0 <= log["*policy_features_sparsity"] < 1.0
    assert log["reward_scale"] == 1.0  # the device-resident synthetic env stack has no reward normaliser
    # --disable_ev: the per-batch moments stay, the explained-variance block goes
    args.disable_ev = True
    log2 = logger.Logger(quiet=True)
    r.log = log2
    r.batch_counter = 3
    r.generate_rollout()
    r.calculate_returns()
    assert "adv_ext_mean" in log2 and "ev_ext" not in log2 and "*policy_features_std" not in log2


def test_tvf_runner_logs_curve_quality():
    """rl/tvf.py:274-301: with TVF on, every 4th batch logs the explained variance of the truncated-value curve
    against fixed n-step Monte-Carlo targets (ev_first / ev_mid / ev_last / ev_average) and *ev_ext."""
    from ppo_amd import value_quality as vq
    args.setup(["--agents=8", "--n_steps=16", "--model_architecture=dual", "--model_encoder=mlp",
                "--model_hidden_units=64", "--env_type=mujoco", "--env_name=Fake", "--seed=9", "--device=cuda",
                "--tvf_enabled=True", "--tvf_value_heads=8", "--tvf_max_horizon=100", "--tvf_return_samples=4",
                "--policy_opt_mini_batch_size=64", "--value_opt_mini_batch_size=64", "--distil_opt_mini_batch_size=64",
                "--env_reward_normalization=off"])
    torch.manual_seed(9)
    np.random.seed(9)
    horizons, weights = tvf.get_value_head_horizons(args.tvf.value_heads, args.tvf.max_horizon, args.tvf.head_spacing,
                                                    include_weight=True)
    model = models.TVFModel("mlp", input_dims=(11,), actions=3, device="cuda", architecture="dual", hidden_units=64,
                            encoder_activation_fn="tanh", tvf_fixed_head_horizons=horizons,
                            tvf_fixed_head_weights=weights, head_scale=0.1, head_bias=True)
    log = logger.Logger(quiet=True)
    r = rollout.Runner(model, log, action_dist="gaussian")
    r.vec_env = FloatVecEnv(8, 11, seed=4)
    r.reset()
    r.batch_counter = 3
    r.generate_rollout()
    r.calculate_returns()
    N, K = r.N, len(horizons)
    est = r.tvf.tvf_value[:N, :, :, 0].cpu().numpy()
    tv = r.tvf.tvf_value[..., 0].cpu().numpy()
    targets = OT.get_return_estimate(mode=args.tvf.return_mode, distribution="fixed", gamma=args.tvf.gamma,
                                     rewards=r.ext_rewards.cpu().numpy(), dones=r.terminals.cpu().numpy(),
                                     required_horizons=np.asarray(horizons), value_sample_horizons=np.asarray(horizons),
                                     value_samples=tv, n_step=args.n_steps, max_samples=args.tvf.return_samples)
    want = Recorder()
    vq.log_curve_quality(want, est, targets, horizons)
    for key in ("ev_first", "ev_mid", "ev_last", "ev_average", "nev_0", "var_1"):
        assert log[key] == pytest.approx(want.got[key], abs=2e-4), key
    assert "*ev_ext" in log and "ev_ext" not in log and "*tvf_return_ext_mean" in log and "*tvf_gamma" in log
    assert "*value_features_sparsity" in log


class Recorder:
    def __init__(self):
        self.got = {}

    def watch_mean(self, key, value, **kw):
        self.got[key] = float(value)


def _rollout_buffers(r):
    torch.cuda.synchronize()
    return {k: getattr(r, k).clone() for k in ("all_obs", "actions", "ext_rewards", "terminals", "value", "log_pac",
                                               "raw_policy")} | {"all_time": torch.from_numpy(r.all_time.copy())}


@pytest.mark.parametrize("stack", ["cartpole_pool", "atari_stack"])
def test_pipelined_rollout_of_gym_api_envs_fills_the_same_buffers_as_the_one_group_loop(stack):
    """north_star: "vectorised envs ... via rl.hybridVecEnv step on host cores with pinned async obs copies".  gym-API envs
    behind the process pool (two worker groups) and the vector wrappers (rms reward normalisation, repeated-action
    penalty) take the Runner's group-pipelined rollout - one group's workers step while the GPU runs the other group's
    policy, uploads are async copies out of the pinned block, the only host wait is the group's own action event - and
    must fill the rollout buffers with the very bytes of the plain forward -> sync -> step loop (rl/rollout.py:730-752),
    over two rollouts (normaliser statistics and penalty counters carry over)."""
    import functools

    def build():
        if stack == "cartpole_pool":
            args.setup(["--agents=16", "--n_steps=24", "--model_architecture=single", "--model_encoder=mlp",
                        "--model_hidden_units=64", "--env_type=classic", "--env_name=CartPole", "--seed=2", "--device=cuda",
                        "--policy_opt_mini_batch_size=128", "--workers=4", "--gamma=0.99", "--disable_logging=True",
                        "--env_max_repeated_actions=3", "--env_repeated_action_penalty=0.25"])
            torch.manual_seed(2)
            np.random.seed(2)
            model = models.TVFModel("mlp", input_dims=(4,), actions=2, device="cuda", architecture="single",
                                    hidden_units=64, head_scale=0.1, head_bias=True)
            r = rollout.Runner(model, logger.Logger(quiet=True))
            r.vec_env = envs.create_envs_classic()
        else:
            args.setup(["--agents=8", "--n_steps=24", "--model_architecture=single", "--model_encoder=impala",
                        "--env_type=atari", "--env_name=FakeGame", "--env_timeout=40", "--env_noop_duration=4", "--seed=5",
                        "--device=cuda", "--policy_opt_mini_batch_size=64", "--workers=8", "--disable_logging=True"])
            # one env per worker: the wrappers draw from the worker's global np.random (as the reference's do), and two
            # envs of one worker step on two threads - their draw order is a race in either rollout form
            torch.manual_seed(5)
            np.random.seed(5)
            shape, nA = envs.get_env_spec()
            model = models.TVFModel("impala", input_dims=shape, actions=nA, device="cuda", architecture="single",
                                    hidden_units=256, head_scale=0.1, head_bias=True)
            r = rollout.Runner(model, logger.Logger(quiet=True))
            r.vec_env = envs.create_envs_classic(
                env_fns=[functools.partial(_fake_atari_env, 100 + 997 * i, args) for i in range(8)])
        return r

    runs = {}
    for mode in ("generic", "pipelined"):
        r = build()
        try:
            from ppo_amd import wrappers
            r.force_generic_rollout = mode == "generic"
            parts = getattr(r.vec_env, "parts", [r.vec_env])
            assert len(parts) == 2 and all(hasattr(p, "step_arrays") for p in parts), "the stack must offer two groups"
            assert wrappers.get_wrapper(r.vec_env, wrappers.VecNormalizeRewardWrapper) is not None
            r.reset()
            bufs = []
            for _ in range(2):
                r.generate_rollout()
                bufs.append(_rollout_buffers(r))
            norm = wrappers.get_wrapper(r.vec_env, wrappers.VecNormalizeRewardWrapper)
            runs[mode] = (bufs, float(norm.ret_rms.var), norm.current_returns.copy(), r.time.copy(), r.ep_count)
        finally:
            r.vec_env.close()
    (ga, gvar, gret, gtime, gep), (pa, pvar, pret, ptime, pep) = runs["generic"], runs["pipelined"]
    for it in range(2):
        for k in ga[it]:
            assert torch.equal(ga[it][k], pa[it][k]), (it, k)
    assert gvar == pvar and np.array_equal(gret, pret) and np.array_equal(gtime, ptime) and gep == pep
    assert int(ga[1]["terminals"].sum()) > 0 and float(ga[1]["ext_rewards"].abs().sum()) > 0
