#!/usr/bin/env python3
"""Generate tests/golden/model_golden.npz by running the REFERENCE's model and PPO update on CPU.

Build container only (needs /root/reference; see ref_shim.py):  python tests/golden/make_model_golden.py

What is captured (data only):
  G4  a seeded reference TVFModel (impala, single architecture, head_scale 0.1, head_bias True,
      rl/config.py:463-468): per-parameter sha256 + moments of its initial state (our initialiser
      must reproduce it bit for bit from the same seed), a uint8 input batch and every forward
      output at policy_temperature 1.0 and 0 (greedy), incl. top-1/top-2 logit margins.
  G5  the reference's Runner.train_policy_minibatch + Runner.optimizer_step (rl/rollout.py:1610-1771,
      1287-1321) driven for 4 optimiser steps on fixed minibatches: loss scalars, gradient norm,
      gradients after the first backward and parameters after steps 1 and 4 (small tensors in full,
      encoder.dense.weight every 16th row).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_shim import load_reference  # noqa: E402

SEED = 1
N_ACTIONS = 6
INPUT_DIMS = (4, 84, 84)
FWD_BATCH = 8
MB = 16
STEPS = 4
DENSE_ROW_STRIDE = 16


def sub(name, t):
    """Full tensor, except the big dense matrix which is sampled by rows."""
    a = t.detach().cpu().numpy()
    if name.endswith("encoder.dense.weight"):
        return a[::DENSE_ROW_STRIDE].copy()
    return a.copy()


def main():
    rl = load_reference([
        "--model_architecture=single", "--model_encoder=impala", "--env_embed_time=False", "--device=cpu",
        "--env_reward_normalization=off", "--disable_ev=True", "--output_folder=/tmp/ref_golden_out",
        f"--agents={MB}", "--n_steps=4", f"--seed={SEED}", f"--policy_opt_mini_batch_size={MB}"])
    import torch
    from rl import config, logger, models, rollout
    args = config.args
    torch.manual_seed(SEED)
    model = models.TVFModel(
        encoder="impala", encoder_args=None, input_dims=INPUT_DIMS, actions=N_ACTIONS, device="cpu",
        architecture="single", dtype=torch.float32, hidden_units=args.model.hidden_units,
        encoder_activation_fn="relu", head_scale=args.model.head_scale, head_bias=args.model.head_bias,
        value_head_names=("ext",))
    out = {}
    meta = {"seed": SEED, "n_actions": N_ACTIONS, "input_dims": INPUT_DIMS, "hidden_units": args.model.hidden_units,
            "head_scale": args.model.head_scale, "head_bias": args.model.head_bias,
            "ppo_epsilon": args.ppo_epsilon, "entropy_bonus": args.entropy_bonus, "ppo_vf_coef": args.ppo_vf_coef,
            "max_grad_norm": args.max_grad_norm, "lr": args.policy_opt.lr, "adam_epsilon": args.policy_opt.adam_epsilon,
            "betas": [args.policy_opt.adam_beta1, args.policy_opt.adam_beta2], "dense_row_stride": DENSE_ROW_STRIDE,
            "params": {}}
    names = [n for n, _ in model.policy_net.named_parameters()]
    for n, p in model.policy_net.named_parameters():
        a = p.detach().numpy()
        meta["params"][n] = {"shape": list(a.shape), "sha256": hashlib.sha256(a.tobytes()).hexdigest(),
                             "sum": float(a.astype(np.float64).sum()), "abs_sum": float(np.abs(a).astype(np.float64).sum())}

    # ---- G4 forward
    rng = np.random.default_rng(SEED)
    x = rng.integers(0, 256, size=(FWD_BATCH, *INPUT_DIMS), dtype=np.uint8)
    out["fwd_x"] = x
    with torch.no_grad():
        r1 = model.forward(x, output="policy", policy_temperature=1.0)
        r0 = model.forward(x, output="policy", policy_temperature=0.0)
    for k in ("raw_policy", "log_policy", "value", "advantage"):
        out[f"fwd_{k}"] = r1[k].numpy()
    out["fwd_greedy_log_policy"] = r0["log_policy"].numpy()
    out["fwd_greedy_argmax_policy"] = r0["argmax_policy"].numpy()
    top2 = np.sort(out["fwd_raw_policy"], axis=1)[:, -2:]
    out["fwd_logit_margin"] = top2[:, 1] - top2[:, 0]
    out["fwd_greedy_actions"] = out["fwd_raw_policy"].argmax(1).astype(np.int64)

    # ---- G5 PPO update through the reference Runner
    log = logger.Logger()
    runner = rollout.Runner(model, log, action_dist="discrete")
    opt = runner.policy_optimizer
    for step in range(STEPS):
        xs = rng.integers(0, 256, size=(MB, *INPUT_DIMS), dtype=np.uint8)
        with torch.no_grad():
            cur = model.forward(xs, output="policy")
        # behaviour policy = current policy perturbed, so ratios straddle the clip range
        old_logits = cur["raw_policy"] + 0.5 * torch.from_numpy(rng.normal(size=(MB, N_ACTIONS)).astype(np.float32))
        old_lp = torch.log_softmax(old_logits, dim=1)
        actions = torch.from_numpy(rng.integers(0, N_ACTIONS, size=(MB,)).astype(np.int64))
        log_pac = old_lp[range(MB), actions]
        adv = torch.from_numpy(rng.normal(size=(MB,)).astype(np.float32))
        ret = torch.from_numpy(rng.normal(size=(MB, 1)).astype(np.float32))
        data = {"prev_state": torch.from_numpy(xs), "actions": actions, "log_policy": old_lp, "log_pac": log_pac,
                "advantages": adv, "returns": ret}
        for k, v in data.items():
            out[f"mb{step}_{k}"] = v.numpy()
        opt.zero_grad(set_to_none=True)
        res = runner.train_policy_minibatch(data, loss_scale=1.0)
        out[f"mb{step}_result"] = np.asarray([res["loss"], res["kl_approx"], res["kl_true"], res["clip_frac"]], np.float64)
        if step == 0:
            for n, p in model.policy_net.named_parameters():
                if p.grad is not None:
                    out["grad0_" + n] = sub(n, p.grad)
                else:
                    meta["params"][n]["grad_none"] = True
        gn = runner.optimizer_step(opt, "policy")
        out[f"mb{step}_grad_norm"] = np.asarray(gn, np.float64)
        if step in (0, STEPS - 1):
            for n, p in model.policy_net.named_parameters():
                out[f"param_after{step + 1}_" + n] = sub(n, p)
    meta["param_names"] = names
    np.savez_compressed(os.path.join(HERE, "model_golden.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "model_golden.json"), "w"), indent=1)
    print("wrote", len(out), "arrays;", sum(v.nbytes for v in out.values()) / 1e6, "MB raw")
    print({k: out[k].tolist() for k in out if k.endswith("_result") or k.endswith("grad_norm")})
    print("margins", out["fwd_logit_margin"])


if __name__ == "__main__":
    main()
