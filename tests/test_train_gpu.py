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
