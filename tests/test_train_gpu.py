"""GPU: the training loop and entry point (rl/ppo.py:47-383, train.py:86-210) end to end on the synthetic env:
`train.main()` writes checkpoints on the reference's schedule into "<output>/<experiment>/<run> [guid]" and a second
invocation with --restore=auto finds that folder, resumes and continues; a Runner restored from a checkpoint
continues BIT-IDENTICALLY to the one that wrote it (model, Adam moments, env counters, host RNG, sampling counter)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--agents=32", "--n_steps=16", "--model_architecture=single", "--model_encoder=impala", "--env_type=synthetic",
         "--env_embed_time=False", "--seed=4", "--policy_opt_mini_batch_size=128", "--policy_opt_epochs=1",
         "--env_warmup_period=5", "--checkpoint_every=1024", "--experiment_name=exp", "--run_name=resume_test"]
BATCH = 32 * 16


def run_main(monkeypatch, out, extra):
    import train
    monkeypatch.setattr(sys, "argv", ["train.py", *FLAGS, f"--output_folder={out}", *extra])
    return train.main()


def test_train_main_checkpoints_then_resumes_with_restore_auto(monkeypatch, tmp_path):
    from ppo_amd import checkpoint
    from ppo_amd.config import args
    out = str(tmp_path)
    r1 = run_main(monkeypatch, out, [f"--epochs={(4 * BATCH - 1) / 1e6}"])
    folder = args.log_folder
    assert os.path.dirname(folder) == os.path.join(out, "exp") and os.path.basename(folder).startswith("resume_test [")
    assert r1.step == 4 * BATCH and r1.net._adam_step == 4 * (BATCH // 128)
    assert os.path.exists(os.path.join(folder, "params.txt")) and os.path.exists(os.path.join(folder, "training_log.csv"))
    cps = r1.get_checkpoints(folder)
    assert cps, "no checkpoint written"
    cp = checkpoint.load(os.path.join(folder, cps[0][1]))
    # schedule (rl/ppo.py:165-172, 334-339): iterations {0, 2, 4}; 2 and 4 are written (same 000M name), 4 last
    assert cp["step"] == 4 * BATCH and cp["world"] == 1 and len(cp["rank_state"]) == 1
    flat_saved = r1.net.flat.clone()
    moments_saved = r1.net.exp_avg.clone()

    # second invocation: finds the folder through its guid, restores, continues to 8 iterations
    r2 = run_main(monkeypatch, out, [f"--epochs={(8 * BATCH - 1) / 1e6}"])
    assert args.log_folder == folder, "restore=auto did not reuse the earlier run's folder"
    # start_iteration = restored_step // batch + 1 = 5 (rl/ppo.py:114): iterations 5, 6, 7 ran
    assert r2.step == 8 * BATCH
    assert r2.net._adam_step == (4 + 3) * (BATCH // 128)
    assert not torch.equal(r2.net.flat, flat_saved) and not torch.equal(r2.net.exp_avg, moments_saved)
    assert torch.isfinite(r2.net.flat).all()

    # --restore=never starts a fresh folder and from step 0
    r3 = run_main(monkeypatch, out, [f"--epochs={(BATCH - 1) / 1e6}", "--restore=never"])
    assert args.log_folder != folder and r3.step == 1 * BATCH and r3.net._adam_step == BATCH // 128
    # --restore=always without a previous run is an error (train.py:146-147)
    monkeypatch.setattr(sys, "argv", ["train.py", *FLAGS, f"--output_folder={out}", "--run_name=never_ran", "--restore=always"])
    import train
    with pytest.raises(SystemExit):
        train.main()


def make_runner(seed, tmp):
    from ppo_amd import envs, logger, models, rollout
    from ppo_amd.config import args
    args.setup([*FLAGS, f"--output_folder={tmp}"])
    torch.manual_seed(seed)
    shape, nA = envs.get_env_spec()
    model = models.TVFModel("impala", input_dims=shape, actions=nA, device="cuda", architecture="single",
                            hidden_units=256, head_scale=0.1, head_bias=True)
    r = rollout.Runner(model, logger.Logger(quiet=True))
    r.vec_env = envs.create_envs_classic()
    r.reset()
    return r


def iteration(r):
    r.generate_rollout()
    r.calculate_returns()
    r.train()


def test_restored_runner_continues_bit_identically(tmp_path):
    np.random.seed(9)
    a = make_runner(1, str(tmp_path))
    for _ in range(2):
        iteration(a)
    path = a.save_checkpoint(str(tmp_path / "checkpoint-000M-params.pt"), a.step)
    assert path.endswith(".pt.gz") and os.path.exists(path)
    iteration(a)
    torch.cuda.synchronize()

    np.random.seed(12345)               # a different host RNG state and different initial weights: all must be restored
    b = make_runner(2, str(tmp_path))
    assert not torch.equal(a.net.flat, b.net.flat)
    assert b.load_checkpoint(str(tmp_path / "checkpoint-000M-params.pt")) == 2 * BATCH
    assert b.net._adam_step == 2 * (BATCH // 128) and b.batch_counter == 2 and b._sample_calls == a._sample_calls - 17
    iteration(b)
    torch.cuda.synchronize()
    assert torch.equal(a.all_obs, b.all_obs), "env state (generator counters) was not restored"
    assert torch.equal(a.actions, b.actions), "sampling counter was not restored"
    assert torch.equal(a.ext_rewards, b.ext_rewards) and torch.equal(a.advantage, b.advantage)
    assert torch.equal(a.net.flat, b.net.flat), "parameters diverged after resume (minibatch permutation RNG?)"
    assert torch.equal(a.net.exp_avg, b.net.exp_avg) and torch.equal(a.net.exp_avg_sq, b.net.exp_avg_sq)
    assert a.step == b.step and a.net._adam_step == b.net._adam_step and a.ep_count == b.ep_count


# ---------------------------------------------------------------- checkpoint structure vs the reference's
def _tree(v):
    """The describe() of tests/golden/make_checkpoint_golden.py, applied to what checkpoint.load returns."""
    if isinstance(v, torch.Tensor):
        return {"__tensor__": str(v.dtype).replace("torch.", ""), "shape": list(v.shape)}
    if isinstance(v, np.ndarray):
        return {"__ndarray__": str(v.dtype), "shape": list(v.shape)}
    if isinstance(v, dict):
        return {"__dict__": {str(k): _tree(x) for k, x in v.items()}, "key_type": sorted({type(k).__name__ for k in v})}
    if isinstance(v, (list, tuple)):
        kinds = [_tree(x) for x in v]
        same = all(k == kinds[0] for k in kinds) if kinds else True
        return {"__seq__": type(v).__name__, "len": len(v), "items": kinds[:1] if same else kinds}
    return {"__scalar__": type(v).__name__}


# reference keys this package does not write, and why: `logs` is the pickled Logger object and `env_state` pickles gym
# wrappers (here: plain per-rank `rank_state[*]['env_state']`); `stats` / `vars` / `discounted_episode_score` are
# logging accumulators of features outside the hot path (reward clipping, crash counting, SNS); the fixture itself was
# taken with logs and env state disabled.
NOT_WRITTEN = {"stats", "vars", "discounted_episode_score"}
CORE_GROUP_KEYS = ("lr", "betas", "eps", "weight_decay", "amsgrad", "maximize", "foreach", "capturable", "params")


def test_checkpoint_key_tree_matches_the_reference(tmp_path, golden_dir):
    """Same top-level keys, same model_state_dict names / shapes, and every optimiser entry in
    torch.optim.Adam.state_dict() layout with state under the same parameter indices, tensor for tensor, as the
    reference's Runner.save_checkpoint produced (rl/rollout.py:394-453; tests/golden/checkpoint_golden.json).  A real
    torch.optim.Adam over parameters of these shapes must accept the optimiser entry."""
    import json
    from ppo_amd import checkpoint
    want = json.load(open(os.path.join(golden_dir, "checkpoint_golden.json")))["impala_single"]
    ref = want["tree"]["__dict__"]
    r = make_runner(3, str(tmp_path))
    iteration(r)
    path = r.save_checkpoint(str(tmp_path / "checkpoint-000M-params.pt"), r.step)
    cp = checkpoint.load(path)
    got = _tree(cp)["__dict__"]
    missing = set(ref) - set(got)
    assert missing == NOT_WRITTEN, missing
    # entries that are the same kind of thing on both sides (per-env arrays differ in the env count only)
    for k in ("step", "ep_count", "batch_counter", "reward_scale"):
        assert got[k] == ref[k], k
    assert got["episode_score"]["__ndarray__"] == ref["episode_score"]["__ndarray__"]
    assert got["model_state_dict"] == ref["model_state_dict"]
    assert list(cp["model_state_dict"])[:39] == ["policy_net." + n for n in want["policy_param_names"]]
    for k in ("policy_optimizer_state_dict", "value_optimizer_state_dict"):
        g, w = got[k]["__dict__"], ref[k]["__dict__"]
        assert set(g) == {"state", "param_groups"}
        gg, wg = g["param_groups"]["items"][0]["__dict__"], w["param_groups"]["items"][0]["__dict__"]
        for key in CORE_GROUP_KEYS:
            assert gg[key] == wg[key], (k, key)
        ours = cp[k]["param_groups"][0]
        for key in ("lr", "eps", "weight_decay", "amsgrad"):
            assert ours[key] == want["param_groups"][k][key]
        assert list(ours["betas"]) == want["param_groups"][k]["betas"] and len(ours["params"]) == want["param_groups"][k]["n_params"]
    # the policy optimiser stepped on both sides: state under the same indices with the same tensors
    assert got["policy_optimizer_state_dict"]["__dict__"]["state"] == ref["policy_optimizer_state_dict"]["__dict__"]["state"]
    assert sorted(cp["policy_optimizer_state_dict"]["state"]) == want["param_groups"]["policy_optimizer_state_dict"]["state_indices"]
    # ... and torch's own Adam takes it
    shapes = [tuple(t.shape) for t in list(cp["model_state_dict"].values())[:39]]
    params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    adam = torch.optim.Adam(params, lr=1.0)
    adam.load_state_dict(cp["policy_optimizer_state_dict"])
    assert adam.param_groups[0]["lr"] == 2.5e-4 and adam.param_groups[0]["eps"] == 1e-5
    o, shape = r.net._offsets["encoder.dense.weight"]
    idx = want["policy_param_names"].index("encoder.dense.weight")
    assert torch.equal(adam.state[params[idx]]["exp_avg"], r.net.exp_avg[o:o + int(np.prod(shape))].view(shape).cpu())
    assert float(adam.state[params[idx]]["step"]) == r.net._adam_step


def test_train_main_under_precision_medium_runs_resumes_and_keeps_the_split_launches(monkeypatch, tmp_path):
    """`train.py --precision=medium` (the reference's default value of the flag, train.py:166-178) end to end: the nets are
    built with the split-bf16 launches, training runs and checkpoints, and a second invocation restores into a net whose
    packed split weights are rebuilt from the restored parameters (the first forward after the restore equals a forward of a
    fresh medium net loaded with the same state: a stale packing would differ)."""
    from ppo_amd import models
    out = str(tmp_path)
    extra = ["--precision=medium", "--run_name=medium_test", "--restore=auto"]
    r1 = run_main(monkeypatch, out, [f"--epochs={(2 * BATCH - 1) / 1e6}", *extra])
    assert r1.net.split_bf16 and r1.net.precision == "medium" and r1.step == 2 * BATCH
    assert torch.isfinite(r1.net.flat).all()
    r2 = run_main(monkeypatch, out, [f"--epochs={(4 * BATCH - 1) / 1e6}", *extra])
    assert r2.net.split_bf16 and r2.step == 4 * BATCH and torch.isfinite(r2.net.flat).all()
    # the packed split weights follow the parameters: a fresh medium net given r2's state computes the same outputs
    x = torch.randint(0, 256, (8, *r2.net.spec.input_dims), dtype=torch.uint8, device="cuda")
    fresh = models.DualHeadNet("impala", r2.net.spec.input_dims, r2.net.n_actions, hidden_units=256, head_scale=0.1, head_bias=True,
                               device="cuda", precision="medium")
    fresh.load_state_dict({k: v.clone() for k, v in r2.net.state_dict().items()})
    a, b = r2.net.forward(x), fresh.forward(x)
    assert torch.equal(a["raw_policy"], b["raw_policy"]) and torch.equal(a["value"], b["value"])
    # and the default stays exact float32
    r3 = run_main(monkeypatch, out, [f"--epochs={(BATCH - 1) / 1e6}", "--run_name=high_test", "--precision=high"])
    assert not r3.net.split_bf16
