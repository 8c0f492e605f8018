"""GPU: the Runner end to end on a small synthetic config — rollout buffers, the fused returns scan
against the oracle, advantage normalisation, and that PPO updates change the policy sensibly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import returns as O  # noqa: E402 (checker)
from ppo_amd import envs, logger, models, rollout  # noqa: E402
from ppo_amd.config import args  # noqa: E402


@pytest.fixture(scope="module")
def runner():
    args.setup(["--agents=16", "--n_steps=32", "--model_architecture=single", "--model_encoder=impala",
                "--env_type=synthetic", "--env_embed_time=False", "--seed=3", "--device=cuda",
                "--policy_opt_mini_batch_size=64", "--policy_opt_epochs=2", "--disable_logging=True"])
    torch.manual_seed(3)
    np.random.seed(3)
    shape, nA = envs.get_env_spec()
    model = models.TVFModel("impala", input_dims=shape, actions=nA, device="cuda", architecture="single",
                            hidden_units=256, head_scale=0.1, head_bias=True)
    r = rollout.Runner(model, logger.Logger(quiet=True))
    r.vec_env = envs.create_envs_classic()
    r.reset()
    return r


def test_rollout_fills_buffers_consistently(runner):
    r = runner
    r.generate_rollout()
    torch.cuda.synchronize()
    N, A = r.N, r.A
    assert r.all_obs.shape == (N + 1, A, 4, 84, 84) and r.all_obs.dtype == torch.uint8
    # the observation stored at t+1 is what the env returned after step t (= current host obs at the end)
    assert np.array_equal(r.all_obs[N].cpu().numpy(), r.obs)
    # log_policy rows are normalised, actions in range, log_pac = log_policy[action]
    lp = r.log_policy.cpu()
    assert torch.allclose(lp.exp().sum(-1), torch.ones(N, A), atol=1e-5)
    act = r.actions.cpu().long()
    assert act.min() >= 0 and act.max() < r.n_actions
    assert torch.equal(r.log_pac.cpu(), lp.gather(2, act[..., None])[..., 0])
    # stored policy/value rows equal a fresh forward on the stored observations
    out = r.model.forward(r.all_obs[5])
    assert torch.allclose(out["log_policy"], r.log_policy[5], atol=1e-6)
    assert torch.allclose(out["value"], r.value[5], atol=1e-6)
    # rewards / terminals made it to the device
    assert abs(float(r.ext_rewards.mean())) < 0.3 and 0.5 < float(r.ext_rewards.std()) < 1.5
    assert r.terminals.dtype == torch.bool
    # actions are spread over the action set (near-uniform initial policy)
    assert len(torch.unique(act)) == r.n_actions


def _ulps(x, ref):
    """Distance in float32 units in the last place (monotone integer view of the bit patterns)."""
    def key(a):
        i = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
        return np.where(i < 0, np.int64(-2**31) - i, i)
    return np.abs(key(x) - key(ref))


def test_returns_match_oracle_within_2_ulp(runner):
    r = runner
    r.calculate_returns()
    torch.cuda.synchronize()
    N, A = r.N, r.A
    v = r.value.view(N + 1, A).cpu().numpy()
    oa, orr = O.gae_and_returns(r.ext_rewards.cpu().numpy(), v[:N], v[N], r.terminals.cpu().numpy(),
                                args.gamma, args.lambda_policy, args.lambda_value)
    a = r.advantage.cpu().numpy()
    b = r.returns.view(N, A).cpu().numpy()
    # A = 16 takes the tiles regime (per-segment affine maps composed in float64, rounded once): not the reference's
    # operation order, so not bit-exact like the columns regime (tests/test_gae_gpu.py), but within 2 ulp of it for
    # bool terminals - and far inside the reference's own 1e-5 criterion
    assert int(_ulps(a, oa).max()) <= 2, int(_ulps(a, oa).max())
    # returns = advantage + value: 2 units in the last place of the larger operand (the sum itself may cancel)
    unit = np.spacing(np.maximum(np.abs(oa), np.abs(v[:N])).astype(np.float32))
    assert (np.abs(b.astype(np.float64) - orr) <= 2.0 * unit).all()
    assert np.abs(a - oa).max() <= 1e-5 * np.abs(oa).max() and np.abs(b - orr).max() <= 1e-5 * np.abs(orr).max()


def test_train_normalises_advantages_and_updates_policy(runner):
    r = runner
    before = r.net.flat.clone()
    r.train()
    torch.cuda.synchronize()
    adv = r.advantage.cpu().numpy().astype(np.float64)
    norm = r.norm_advantage.cpu().numpy()
    ref = (adv - adv.mean()) / (adv.std() + args.advantage_epsilon)
    assert np.abs(norm - ref).max() < 1e-5 * np.abs(ref).max()
    stats = r.fetch_stats()
    n_updates = 2 * (r.N * r.A // 64)
    assert r.net._adam_step == n_updates
    delta = (r.net.flat - before).abs()
    assert float(delta.max()) > 0 and float(delta.max()) < 2.5e-4 * n_updates * 1.01  # Adam moves <= lr per step
    assert 0.0 <= stats["clip_frac"] <= 1.0 and stats["entropy"] > 1.0 and np.isfinite(stats["loss_policy"])
    assert stats["grad_policy"] > 0 and abs(stats["adv_mean"] - adv.mean()) < 1e-5
    # advantage head and log_std never receive gradient (rl/models.py:506; reference grad is None)
    assert torch.equal(r.net.params["advantage_head.weight"], before[r.net._offsets["advantage_head.weight"][0]:][:6 * 256].view(6, 256))


def test_pipelined_rollout_equals_unsplit_rollout(runner):
    """Splitting the envs into two groups (host stepping of one overlaps the GPU policy step of the other)
    must not change a single byte of the rollout: env streams and the sampling counter are keyed by the
    global env index.  Runs on the module's model; A=32 so that each group has 16 envs."""
    from ppo_amd.vec_env import SplitVecEnv, SyntheticVecEnv
    old = (args.agents, args.n_steps)
    args.agents, args.n_steps = 32, 8
    try:
        outs = []
        for parts in (1, 2):
            r = rollout.Runner(runner.model, logger.Logger(quiet=True))
            per = 32 // parts
            envs_ = [SyntheticVecEnv(per, seed=3, p_done=0.05, env_offset=i * per, threads=2) for i in range(parts)]
            r.vec_env = envs_[0] if parts == 1 else SplitVecEnv(envs_)
            r.reset()
            r.generate_rollout()
            r.generate_rollout()  # second rollout: the sampling counter advanced identically
            torch.cuda.synchronize()
            outs.append([x.cpu().clone() for x in (r.all_obs, r.actions, r.log_policy, r.value, r.ext_rewards,
                                                   r.terminals, r.log_pac)] + [torch.from_numpy(np.array(r.obs))])
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        assert outs[0][5].any() and len(torch.unique(outs[0][1])) > 1
    finally:
        args.agents, args.n_steps = old


def test_split_chain_timeout_is_survived_and_the_rollout_redone(runner, monkeypatch):
    """The two-workgroups-per-image chained launch (models.CHAIN_SPLIT, default on for rollout groups <= 128 images)
    spin-waits on a partner workgroup with a bounded spin.  Drive the timeout on purpose (control word 3: every second
    workgroup withholds its flags): generate_rollout must not raise - it drops to the one-workgroup launch, puts the
    (exactly restorable) synthetic env back to the rollout's start and redoes it, and the buffers of that and of the
    following rollout are byte-identical to a run that never used the split launch."""
    from ppo_amd import models
    from ppo_amd.vec_env import SplitVecEnv, SyntheticVecEnv
    old = (args.agents, args.n_steps)
    args.agents, args.n_steps = 32, 6
    try:
        outs, warned = [], []
        for split in (0, 1):
            monkeypatch.setattr(models, "CHAIN_SPLIT", split)
            r = rollout.Runner(runner.model, logger.Logger(quiet=True))
            net = r.policy_net
            net._chain_split_usable, net._plans = None, {}
            r.log.warn = lambda msg, _w=warned: _w.append(msg)
            r.vec_env = SplitVecEnv([SyntheticVecEnv(16, seed=5, p_done=0.05, env_offset=i * 16, threads=2) for i in range(2)])
            r.reset()
            r.generate_rollout()
            if split:
                calls = []
                orig = net._call
                net._call = lambda fn, *a: (calls.append(fn), orig(fn, *a))[1]
                assert net._chain_split_usable is True and not net.chain_split_error()
                net.chain_split_inject_fault()
                ep_before = r.ep_count
            r.generate_rollout()  # with split: times out, is noticed, redone on the one-workgroup launch
            torch.cuda.synchronize()
            snap = [x.cpu().clone() for x in (r.all_obs, r.actions, r.log_policy, r.raw_policy, r.value, r.ext_rewards,
                                              r.terminals, r.log_pac)] + [torch.from_numpy(np.array(r.obs)),
                                                                          torch.from_numpy(r.all_time.copy())]
            if split:
                assert net._chain_split_usable is False and not net.chain_split_error()
                assert len(warned) == 1 and "redoing this rollout from its start state" in warned[0]
                assert r.ep_count - ep_before == int(r.terminals.sum())  # episodes of the void attempt are not counted twice
            r.generate_rollout()  # and the run goes on: no split launch any more
            torch.cuda.synchronize()
            if split:
                assert "ppo_impala_stack_chain_split_forward_f32" not in calls[-40:]
            outs.append(snap + [r.all_obs.cpu().clone(), r.actions.cpu().clone(), r.value.cpu().clone()])
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        assert outs[0][6].any()
    finally:
        args.agents, args.n_steps = old
        runner.policy_net._chain_split_usable, runner.policy_net._plans = None, {}


def test_second_iteration_runs_and_checkpoint_round_trips(runner, tmp_path):
    r = runner
    r.generate_rollout()
    r.calculate_returns()
    r.train()
    torch.cuda.synchronize()
    assert torch.isfinite(r.net.flat).all()
    path = str(tmp_path / "checkpoint-001M-params.pt")
    r.save_checkpoint(path, 12345)
    w = r.net.flat.clone()
    m = r.net.exp_avg.clone()
    r.net.flat.zero_()
    r.net.exp_avg.zero_()
    assert r.load_checkpoint(path) == 12345
    assert torch.equal(r.net.flat, w) and torch.equal(r.net.exp_avg, m)
