"""GPU: micro-batch gradient accumulation (rl/rollout.py:2257-2407, 2331-2374).  The reference splits a minibatch into
micro-batches of at most --max_micro_batch_size samples, scales each pass by 1 / micro_batches and lets autograd
accumulate; here every backward pass overwrites the flat gradient, so passes are summed by ppo_accumulate_f32.  A split
minibatch must give the update of the unsplit one up to float32 summation order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from ppo_amd import envs, logger, models, rollout  # noqa: E402
from ppo_amd.config import args  # noqa: E402

FLAGS = ["--agents=16", "--n_steps=16", "--model_architecture=single", "--model_encoder=impala", "--env_type=synthetic",
         "--env_embed_time=False", "--seed=6", "--device=cuda", "--policy_opt_mini_batch_size=128",
         "--policy_opt_epochs=1", "--disable_logging=True"]


def make(extra=()):
    args.setup([*FLAGS, *extra])
    torch.manual_seed(6)
    shape, nA = envs.get_env_spec()
    model = models.TVFModel("impala", input_dims=shape, actions=nA, device="cuda", architecture="single",
                            hidden_units=256, head_scale=0.1, head_bias=True)
    r = rollout.Runner(model, logger.Logger(quiet=True))
    r.vec_env = envs.create_envs_classic()
    r.reset()
    np.random.seed(6)
    r.generate_rollout()
    r.calculate_returns()
    return r


def batch_of(r):
    B = r.N * r.A
    r._normalize_advantages()
    return {"prev_state": r.all_obs[:r.N].reshape(B, *r.state_shape), "actions": r.actions.reshape(B).long(),
            "log_policy": r.log_policy.reshape(B, -1), "log_pac": r.log_pac.reshape(B),
            "advantages": r.norm_advantage.reshape(B), "returns": r.returns.reshape(B, 1)}


def rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def test_accumulated_gradient_equals_the_one_pass_gradient_before_adam():
    """The quantity itself, before any optimiser amplifies it: the gradient buffer after the 4 micro-batch passes of a
    minibatch (summed by ppo_accumulate_f32) against the gradient of the same 128 samples in one pass — float32
    round-off of a different summation order, nothing else.  An `after_mini_batch` hook that stops the epoch leaves the
    buffer as the optimiser step would have seen it."""
    grads = []
    for micro in (None, 32):
        r = make()
        np.random.seed(77)
        ctx = r.train_batch(batch_of(r), r.train_policy_minibatch, 128, r.policy_optimizer, "policy", epoch=0,
                            force_micro_batch_size=micro, hooks={"after_mini_batch": lambda c: True})
        assert ctx["did_break"] and len(ctx["outputs"]) == (1 if micro is None else 4) and r.net._adam_step == 0
        torch.cuda.synchronize()
        grads.append({n: g.clone() for n, g in r.net.grads.items()})
        flat = r.net.grad.clone()
        grads.append(flat)
    (one, one_flat), (acc, acc_flat) = (grads[0], grads[1]), (grads[2], grads[3])
    gmax = float(one_flat.abs().max())
    assert gmax > 1e-3
    assert float((acc_flat - one_flat).abs().max()) <= 2e-6 * gmax, "accumulated gradient is not the one-pass gradient"
    for n in one:
        tmax = float(one[n].abs().max())
        if tmax > 0:
            assert float((acc[n] - one[n]).abs().max()) <= 1e-5 * tmax, n
        else:
            assert float(acc[n].abs().max()) == 0.0, n


def test_train_batch_micro_batches_accumulate_to_the_same_update():
    ref = make()
    np.random.seed(77)
    calls = {"micro": 0, "mini": 0}
    ctx = ref.train_batch(batch_of(ref), ref.train_policy_minibatch, 128, ref.policy_optimizer, "policy", epoch=0,
                          hooks={"after_micro_batch": lambda c: calls.__setitem__("micro", calls["micro"] + 1),
                                 "after_mini_batch": lambda c: calls.__setitem__("mini", calls["mini"] + 1)})
    assert ctx["mini_batches"] == 2 and len(ctx["outputs"]) == 2 and calls == {"micro": 2, "mini": 2}
    assert set(ctx["outputs"][0]) >= {"loss", "kl_approx", "clip_frac"} and ref.net._adam_step == 2

    split = make()
    assert torch.equal(split.net.flat, make().net.flat)  # same start
    np.random.seed(77)
    seen = []
    ctx = split.train_batch(batch_of(split), split.train_policy_minibatch, 128, split.policy_optimizer, "policy", epoch=0,
                            force_micro_batch_size=32, hooks={"after_micro_batch": lambda c: seen.append(dict(c))})
    assert ctx["mini_batches"] == 2 and len(ctx["outputs"]) == 8 and split.net._adam_step == 2
    assert [c["micro_batch"] for c in seen] == [0, 1, 2, 3, 0, 1, 2, 3] and seen[0]["is_first"] and seen[-1]["is_last"]
    torch.cuda.synchronize()
    # Adam's first steps move a weight by ~lr * g / (|g| + eps): where |g| ~ eps = 1e-5 a rounding difference of the
    # gradient (see the test above: equal to float32 round-off, ~1e-8 absolute) is amplified to a fraction of a
    # learning-rate step.  So: a loose bar on those weights, a tight one wherever the gradient is well above eps
    # (|exp_avg| after two steps ~ 0.19 |g|), and the moments themselves to round-off.
    lr = 2.5e-4
    d = (split.net.flat - ref.net.flat).abs()
    assert float(d.max()) < 0.25 * 2 * lr and float(d.median()) < 1e-3 * lr
    clear = ref.net.exp_avg.abs() > 2e-4  # |g| >~ 1e-3 = 100 eps
    assert int(clear.sum()) > 1000
    assert float(d[clear].max()) < 1e-3 * lr, "a weight with a clear gradient moved differently"
    # (the second step's gradient is taken at weights that already differ by those amplified roundings)
    assert rel(split.net.exp_avg, ref.net.exp_avg) < 5e-5, "accumulated gradient differs from the one-pass gradient"

    # a hook that stops the epoch: no optimiser step for that minibatch
    stop = make()
    ctx = stop.train_batch(batch_of(stop), stop.train_policy_minibatch, 128, stop.policy_optimizer, "policy",
                           hooks={"after_mini_batch": lambda c: True})
    assert ctx.get("did_break") is True and ctx["mini_batches"] == 1 and stop.net._adam_step == 0
    with pytest.raises(Exception, match="Not supported"):
        stop.train_batch(batch_of(stop), stop.train_policy_minibatch, 128, stop.policy_optimizer, "policy", delta_threshold=0.1)
    with pytest.raises(ValueError):
        stop.train_batch(batch_of(stop), stop.train_policy_minibatch, 128, stop.policy_optimizer, "policy", force_micro_batch_size=48)
    # thinning: a fraction of every micro-batch
    sizes = []
    stop.train_batch(batch_of(stop), lambda d, loss_scale: sizes.append(len(d["prev_state"])) or stop.train_policy_minibatch(d, loss_scale=loss_scale),
                     128, stop.policy_optimizer, "policy", thinning=0.5)
    assert sizes == [64, 64]


def test_max_micro_batch_size_flag_splits_the_fast_path():
    one = make()
    np.random.seed(5)
    one.train()
    two = make(["--max_micro_batch_size=64"])
    np.random.seed(5)
    two.train()
    torch.cuda.synchronize()
    assert one.net._adam_step == two.net._adam_step == 2
    assert rel(two.net.exp_avg, one.net.exp_avg) < 1e-4
    a, b = one.fetch_stats(), two.fetch_stats()
    for k in ("loss_policy", "entropy", "kl_approx", "clip_frac"):
        assert abs(a[k] - b[k]) <= 1e-5 * max(1.0, abs(a[k])), k


def test_minibatch_read_through_the_permutation_equals_the_gathered_minibatch():
    """uint8 image minibatches are read out of the whole batch by the first convolution and by its weight gradient
    (ppo_conv3x3_pool_forward_packed_indexed_f32 / ..._slabs_pooled_indexed_f32) instead of from a gathered copy: same
    statistics, same gradient, bit for bit."""
    from ppo_amd import models
    torch.manual_seed(5)
    net = models.DualHeadNet("impala", (4, 84, 84), 6, hidden_units=256, head_scale=0.1, head_bias=True, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    Btot, B = 96, 32
    obs = torch.randint(0, 256, (Btot, 4, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
    idx = torch.randperm(Btot, device="cuda", generator=g)[:B].int().contiguous()
    actions = torch.randint(0, 6, (Btot,), dtype=torch.int32, device="cuda", generator=g)
    adv = torch.randn(Btot, device="cuda", generator=g)
    ret = torch.randn(Btot, 1, device="cuda", generator=g)
    lp = torch.log_softmax(torch.randn(Btot, 6, device="cuda", generator=g), dim=1)
    pac = lp.gather(1, actions.long()[:, None])[:, 0].contiguous()
    assert net.takes_obs_index(obs)
    gathered = obs[idx.long()].contiguous()
    net.grad.zero_()
    s_g = net.ppo_minibatch(gathered, actions, pac, lp, adv, ret, index=idx).clone()
    g_g = net.grad.clone()
    calls = []
    orig = net._call
    net._call = lambda fn, *a: (calls.append(fn), orig(fn, *a))[1]
    net.grad.zero_()
    s_i = net.ppo_minibatch(obs, actions, pac, lp, adv, ret, index=idx).clone()
    torch.cuda.synchronize()
    assert "ppo_conv3x3_pool_forward_packed_indexed_f32" in calls and "ppo_conv3x3_backward_weight_slabs_pooled_indexed_f32" in calls
    assert torch.equal(s_g, s_i) and torch.equal(g_g, net.grad) and float(g_g.abs().max()) > 0
    assert net.obs_index is None
