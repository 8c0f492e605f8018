"""CPU: the host-side env layer — reward normaliser / repeated-action penalty /
RunningMeanStd bit-exact against fixtures recorded from the reference's own classes
(tests/golden/make_wrappers_golden.py), and the process-pool vector env's contract
(rl/hybridVecEnv.py:49-203): ordering, determinism vs in-process envs, auto-reset, action -1,
seed / save_state / restore_state through the workers, and loud worker failures."""
import functools
import os

import numpy as np
import pytest

from ppo_amd import classic_envs, wrappers
from ppo_amd.hybrid_vec_env import HybridAsyncVectorEnv
from ppo_amd.running_stats import RunningMeanStd

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "wrappers_golden.npz"))


class ScriptedVecEnv:
    def __init__(self):
        self.num_envs = GOLD["rewards"].shape[1]
        self.t = 0

    def reset(self):
        self.t = 0
        return np.zeros((self.num_envs, 1), np.float32)

    def step(self, a):
        t = self.t
        self.t += 1
        return (np.zeros((self.num_envs, 1), np.float32), GOLD["rewards"][t].copy(), GOLD["dones"][t].copy(),
                [{"time": t} for _ in range(self.num_envs)])


@pytest.mark.parametrize("tag,kw", [("rms", dict(gamma=0.999, clip=10.0)),
                                    ("rms_scale", dict(gamma=0.99, clip=2.0, scale=0.5)),
                                    ("ema", dict(gamma=0.999, clip=-1, mode="ema"))])
def test_reward_normaliser_matches_reference_bitwise(tag, kw):
    w = wrappers.VecNormalizeRewardWrapper(ScriptedVecEnv(), **kw)
    w.reset()
    for t in range(GOLD["rewards"].shape[0]):
        _, r, _, infos = w.step(GOLD["actions"][t])
        want = GOLD[f"norm_{tag}_rewards"][t]
        assert r.dtype == want.dtype and np.array_equal(r, want), t
        assert w.std == GOLD[f"norm_{tag}_std"][t]
        assert int(infos[0].get("reward_clips", 0)) == GOLD[f"norm_{tag}_clips"][t]
    final = np.asarray([w.ret_rms.mean, w.ret_rms.var, w.ret_rms.count, w.ret_var], np.float64)
    assert np.array_equal(final, GOLD[f"norm_{tag}_final"])
    assert np.array_equal(np.asarray(w.current_returns), GOLD[f"norm_{tag}_current_returns"])
    assert GOLD["norm_rms_scale_clips"].sum() > 0  # the fixture does exercise the clip


def test_reward_normaliser_state_round_trip_and_sync_hook():
    w = wrappers.VecNormalizeRewardWrapper(ScriptedVecEnv(), gamma=0.999)
    w.reset()
    for t in range(20):
        w.step(GOLD["actions"][t])
    buf = {}
    w.save_state(buf)
    w2 = wrappers.VecNormalizeRewardWrapper(ScriptedVecEnv(), gamma=0.999)
    w2.restore_state(buf)
    assert w2.std == w.std and np.array_equal(w2.current_returns, w.current_returns)
    assert wrappers.get_wrapper(wrappers.VecRepeatedActionPenalty(w, 3), wrappers.VecNormalizeRewardWrapper) is w
    # moments_sync = identity must reproduce the unsynchronised normaliser to rounding
    w3 = wrappers.VecNormalizeRewardWrapper(ScriptedVecEnv(), gamma=0.999, moments_sync=lambda m: m)
    w3.reset()
    for t in range(20):
        w3.step(GOLD["actions"][t])
    assert abs(w3.std - w.std) < 1e-9 * w.std


def test_repeated_action_penalty_matches_reference():
    w = wrappers.VecRepeatedActionPenalty(ScriptedVecEnv(), max_repeated_actions=5, penalty=0.25)
    w.reset()
    for t in range(GOLD["rewards"].shape[0]):
        _, r, _, infos = w.step(GOLD["actions"][t])
        assert np.array_equal(r, GOLD["penalty_rewards"][t]), t
        assert int(infos[0]["max_repeats"]) == GOLD["penalty_max_repeats"][t]
        assert [int("repeated_action" in i) for i in infos] == list(GOLD["penalty_flagged"][t])
    assert GOLD["penalty_flagged"].sum() > 0


def test_running_mean_std_matches_reference_bitwise():
    rms = RunningMeanStd(shape=(3,))
    start = 0
    for k, n in enumerate(GOLD["rms_batch_sizes"]):
        rms.update(GOLD["rms_batches"][start:start + n])
        start += n
        assert np.array_equal(np.concatenate([rms.mean, rms.var, [rms.count]]), GOLD["rms_trace"][k])
    state = rms.save_state()
    other = RunningMeanStd(shape=(3,))
    other.restore_state(state)
    assert np.array_equal(other.mean, rms.mean) and other.count == rms.count


# ------------------------------------------------------------------------------------------- process pool
def _fns(n, base_seed=100):
    return [functools.partial(classic_envs.make_cartpole, base_seed + i * 997) for i in range(n)]


@pytest.fixture(scope="module")
def pool():
    env = HybridAsyncVectorEnv(_fns(8), max_cpus=2)
    yield env
    env.close()


def test_cartpole_physics_and_limits():
    env = classic_envs.CartPoleEnv(seed=0)
    obs = env.reset()
    assert obs.shape == (4,) and obs.dtype == np.float32 and np.abs(obs).max() <= 0.05
    # pushing right forever tips the pole over the 12 degree limit within a few dozen steps
    for t in range(200):
        obs, r, done, _ = env.step(1)
        assert r == 1.0
        if done:
            break
    assert done and 5 < t < 60 and (abs(obs[2]) > classic_envs.CartPoleEnv.THETA_LIMIT or abs(obs[0]) > 2.4)
    # known answer from rest: theta_acc = -(F/m) / (l (4/3 - m_p/m)), x_acc = F/m - m_p l theta_acc / m
    env._s = np.zeros(4)
    env.step(1)
    th_acc = -(10.0 / 1.1) / (0.5 * (4.0 / 3.0 - 0.1 / 1.1))
    x_acc = 10.0 / 1.1 - 0.05 * th_acc / 1.1
    assert np.allclose(env._s, [0.0, 0.02 * x_acc, 0.0, 0.02 * th_acc], rtol=1e-12) and abs(env._s[1] - 0.19512195) < 1e-8


def test_pool_matches_in_process_envs_and_orders_envs(pool):
    local = [fn() for fn in _fns(8)]
    want = np.stack([e.reset() for e in local])
    obs = pool.reset()
    assert obs.shape == (8, 4) and obs.dtype == np.float32 and np.array_equal(obs, want)
    rng = np.random.default_rng(0)
    finished = 0
    for t in range(120):
        a = rng.integers(0, 2, size=8)
        obs, rew, done, infos = pool.step(a)
        assert rew.dtype == np.float32 and done.dtype == bool and len(infos) == 8
        for i, e in enumerate(local):
            o, r, d, info = e.step(a[i])
            if d:
                o = e.reset()  # the pool auto-resets and returns the first observation of the next episode
            assert np.array_equal(obs[i], o) and rew[i] == r and done[i] == d
            assert infos[i]["ep_length"] == info["ep_length"] and infos[i]["ep_score"] == info["ep_score"]
            finished += d
    assert finished >= 8  # random play ends episodes within ~20 steps


def test_pool_null_action_freezes_env(pool):
    pool.reset()
    a = np.zeros(8, np.int64)
    obs0, _, _, infos0 = pool.step(a)
    a[[1, 6]] = -1
    obs1, rew1, done1, infos1 = pool.step(a)
    for i in (1, 6):
        assert np.array_equal(obs1[i], obs0[i]) and rew1[i] == 0 and not done1[i]
        assert infos1[i]["ep_length"] == infos0[i]["ep_length"]
    assert not np.array_equal(obs1[0], obs0[0]) and rew1[0] == 1


def test_pool_seed_and_state_round_trip(pool):
    pool.seed([7 + i for i in range(8)])
    first = pool.reset()
    pool.seed([7 + i for i in range(8)])
    assert np.array_equal(pool.reset(), first)
    for _ in range(3):
        pool.step(np.ones(8, np.int64))
    saved = {}
    pool.save_state(saved)
    assert sorted(saved) == [f"vec_{i:03d}" for i in range(8)]
    a = np.array([0, 1] * 4)
    after = [pool.step(a)[0].copy() for _ in range(4)]
    pool.restore_state(saved)
    again = [pool.step(a)[0].copy() for _ in range(4)]
    assert all(np.array_equal(x, y) for x, y in zip(after, again))


def test_pool_rejects_bad_sizes_and_reports_worker_errors():
    with pytest.raises(AssertionError, match="must be a multiple of the CPU count"):
        HybridAsyncVectorEnv(_fns(3), max_cpus=2)
    env = HybridAsyncVectorEnv(_fns(2), max_cpus=2, copy=False)
    try:
        assert env.reset() is env.obs  # copy=False hands out the shared block itself
        with pytest.raises(ValueError):
            env.step(np.zeros(3, np.int64))
        with pytest.raises(RuntimeError, match="env worker 0 failed during `load`"):
            env.restore_state({"vec_000": {"nope": 1}, "vec_001": {"nope": 1}})
    finally:
        env.close()


def test_a_wrapper_that_only_overrides_step_is_never_bypassed_by_group_stepping():
    """The Runner's pipelined rollout steps `env.parts` groups directly.  A VecWrapper subclass that defines neither
    `parts` nor the group hooks must not inherit them from the env it wraps (its step() would never run): the group
    interface is not forwarded, so the Runner sees one whole env and takes the generic path."""
    class Inner:
        num_envs = 4
        parts = ["a group"]
        exact_snapshot = True

        def step_arrays(self, *a):
            raise AssertionError("stepped behind the wrapper's back")

        def finish_rollout(self, *a):
            raise AssertionError

        def something_else(self):
            return 7

    class OnlyStep(wrappers.VecWrapper):
        def step(self, actions):
            return "mine"

    w = OnlyStep(Inner())
    assert w.something_else() == 7 and w.num_envs == 4  # ordinary attributes still reach the wrapped env
    for name in ("parts", "step_arrays", "finish_rollout", "step_upload", "leaves", "obs_t"):
        assert not hasattr(w, name), name
    assert getattr(w, "parts", [w]) == [w] and not w.exact_snapshot
    # the two in-tree wrappers define the group interface themselves and can be put back exactly over an exact env
    pen = wrappers.VecRepeatedActionPenalty(Inner(), 3)
    assert "parts" in type(pen).__dict__ and pen.exact_snapshot
    state = pen.snapshot_state()
    pen.prev_actions[:] = 5
    pen.duplicate_counter[:] = 9
    pen.restore_snapshot(state)
    assert (pen.prev_actions == 0).all() and (pen.duplicate_counter == 0).all()
