#!/usr/bin/env python3
"""Generate tests/golden/variants_golden.npz: the REFERENCE's model and Runner on CPU for the configurations
beyond IMPALA/single/discrete — MLP encoder, tanh activation, dual architecture (policy / value / distil
phases), TVF value heads and gaussian policies.

Build container only (needs /root/reference; see ref_shim.py):  python tests/golden/make_variants_golden.py

Variants (each: seeded init hashes, forward outputs, one minibatch of every phase with losses + gradients):
  mlp_gauss_tvf   mlp / tanh / dual / gaussian (3 actions) / 8 TVF heads / obs (11,) / hidden 64   [C5-shaped]
  mlp_disc        mlp / relu / dual / discrete (2 actions) / obs (4,) / hidden 64                   [C1-shaped + DNA]
  humanoid        mlp / tanh / dual / gaussian (17 actions) / 128 TVF heads (max horizon 30000: the geometric
                  spacing keeps the distinct ones) / obs (377,) / hidden 256      [BASELINE configs[4] at its real size]
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_shim import load_reference  # noqa: E402

MB = 32
out, meta = {}, {}


def grads_of(net):
    return {n: (None if p.grad is None else p.grad.detach().numpy().copy()) for n, p in net.named_parameters()}


def record_grads(prefix, net, m):
    for n, g in grads_of(net).items():
        if g is None:
            m.setdefault("grad_none", {}).setdefault(prefix, []).append(n)
        else:
            out[f"{prefix}_grad_{n}"] = g


def run_variant(tag, flags, input_dims, n_actions, action_dist, activation, use_tvf, hidden):
    # the reference's config is a module singleton that can be set up once: one process per variant
    rl = load_reference(flags)
    import torch
    from rl import config, logger, models, rollout
    import rl.tvf
    args = config.args
    m = meta.setdefault(tag, {})
    torch.manual_seed(7)
    horizons = weights = None
    if use_tvf:
        horizons, weights = rl.tvf.get_value_head_horizons(args.tvf.value_heads, args.tvf.max_horizon,
                                                           args.tvf.head_spacing, include_weight=True)
        args.tvf.value_heads = len(horizons)
        out[f"{tag}_tvf_horizons"] = np.asarray(horizons)
        out[f"{tag}_tvf_weights"] = np.asarray(weights, np.float32)
    model = models.TVFModel(
        encoder="mlp", encoder_args=None, input_dims=input_dims, actions=n_actions, device="cpu",
        architecture="dual", dtype=torch.float32, hidden_units=hidden, encoder_activation_fn=activation,
        tvf_fixed_head_horizons=horizons, tvf_fixed_head_weights=weights, head_scale=args.model.head_scale,
        head_bias=args.model.head_bias, value_head_names=("ext",))
    m.update(input_dims=list(input_dims), n_actions=n_actions, action_dist=action_dist, activation=activation,
             hidden=hidden, head_scale=args.model.head_scale, head_bias=args.model.head_bias,
             ppo_epsilon=args.ppo_epsilon, entropy_bonus=args.entropy_bonus, ppo_vf_coef=args.ppo_vf_coef,
             tvf_coef=args.tvf.coef, distil_beta=args.distil.beta, params={})
    for prefix, net in (("policy_net", model.policy_net), ("value_net", model.value_net)):
        for n, p in net.named_parameters():
            a = p.detach().numpy()
            m["params"][f"{prefix}.{n}"] = {"shape": list(a.shape), "sha256": hashlib.sha256(a.tobytes()).hexdigest()}
    m["state_dict_keys"] = list(model.state_dict().keys())

    rng = np.random.default_rng(11)
    x = rng.standard_normal((MB, *input_dims)).astype(np.float32)
    out[f"{tag}_x"] = x
    with torch.no_grad():
        for mode in ("default", "full", "policy", "value"):
            r = model.forward(x, output=mode)
            m.setdefault("forward_keys", {})[mode] = sorted(r.keys())
            for k, v in r.items():
                out[f"{tag}_fwd_{mode}_{k}"] = v.numpy()

    runner = rollout.Runner(model, logger.Logger(), action_dist=action_dist)
    # perturb log_std away from 0 so its gradient path is exercised
    if action_dist == "gaussian":
        with torch.no_grad():
            model.policy_net.log_std.copy_(torch.from_numpy(rng.normal(scale=0.3, size=n_actions).astype(np.float32)))
        out[f"{tag}_log_std"] = model.policy_net.log_std.detach().numpy().copy()

    adv = torch.from_numpy(rng.normal(size=(MB,)).astype(np.float32))
    ret = torch.from_numpy(rng.normal(size=(MB, 1)).astype(np.float32))
    xt = torch.from_numpy(x)
    with torch.no_grad():
        cur = model.forward(x, output="policy")
    # ---- policy phase
    if action_dist == "gaussian":
        std = torch.exp(model.policy_net.log_std.detach())
        old_mu = cur["raw_policy"] + 0.3 * torch.from_numpy(rng.normal(size=(MB, n_actions)).astype(np.float32))
        actions = old_mu + std * torch.from_numpy(rng.normal(size=(MB, n_actions)).astype(np.float32))
        log_pac = torch.distributions.normal.Normal(old_mu, std).log_prob(actions)
        data = {"prev_state": xt, "actions": actions, "log_pac": log_pac, "advantages": adv}
    else:
        old_lp = torch.log_softmax(cur["raw_policy"] + 0.5 * torch.from_numpy(rng.normal(size=(MB, n_actions)).astype(np.float32)), dim=1)
        actions = torch.from_numpy(rng.integers(0, n_actions, size=(MB,)).astype(np.int64))
        data = {"prev_state": xt, "actions": actions, "log_policy": old_lp, "log_pac": old_lp[range(MB), actions],
                "advantages": adv}
    for k, v in data.items():
        out[f"{tag}_policy_{k}"] = v.numpy()
    runner.policy_optimizer.zero_grad(set_to_none=True)
    res = runner.train_policy_minibatch(data, loss_scale=1.0)
    out[f"{tag}_policy_result"] = np.asarray([res["loss"], res["kl_approx"], res["kl_true"], res["clip_frac"]], np.float64)
    record_grads(f"{tag}_policy", model.policy_net, m)

    # ---- value phase (value_net)
    vdata = {"prev_state": xt, "returns": ret}
    if use_tvf:
        K = len(horizons)
        vdata["tvf_returns"] = torch.from_numpy(rng.normal(size=(MB, K)).astype(np.float32))
    for k, v in vdata.items():
        out[f"{tag}_value_{k}"] = v.numpy()
    runner.value_optimizer.zero_grad(set_to_none=True)
    res = runner.train_value_minibatch(vdata, loss_scale=1.0)
    out[f"{tag}_value_result"] = np.asarray([res["loss"], res["loss_std"]], np.float64)
    record_grads(f"{tag}_value", model.value_net, m)
    if use_tvf:
        # the same minibatch with --tvf_head_weighting=h_weighted (rl/tvf.py:55-62)
        args.tvf.head_weighting = "h_weighted"
        runner.value_optimizer.zero_grad(set_to_none=True)
        res = runner.train_value_minibatch(vdata, loss_scale=1.0)
        out[f"{tag}_valuehw_result"] = np.asarray([res["loss"], res["loss_std"]], np.float64)
        record_grads(f"{tag}_valuehw", model.value_net, m)
        args.tvf.head_weighting = "off"

    # ---- distil phase (policy_net learns value_net's estimates under a policy constraint)
    ddata = {"prev_state": xt}
    if use_tvf:
        ddata["distil_targets"] = torch.from_numpy(rng.normal(size=(MB, len(horizons))).astype(np.float32))
    else:
        ddata["distil_targets"] = torch.from_numpy(rng.normal(size=(MB,)).astype(np.float32))
    with torch.no_grad():
        cur = model.forward(x, output="policy")
    noise = 0.2 * torch.from_numpy(rng.normal(size=(MB, n_actions)).astype(np.float32))
    ddata["old_raw_policy"] = cur["raw_policy"] + noise
    ddata["old_log_policy"] = torch.log_softmax(cur["raw_policy"] + noise, dim=1)
    for k, v in ddata.items():
        out[f"{tag}_distil_{k}"] = v.numpy()
    opt = runner.distil_optimizer
    opt.zero_grad(set_to_none=True)
    for p in model.policy_net.parameters():
        p.grad = None
    res = runner.train_distil_minibatch(ddata, loss_scale=1.0)
    out[f"{tag}_distil_result"] = np.asarray([res["loss"], res["loss_std"]], np.float64)
    record_grads(f"{tag}_distil", model.policy_net, m)


VARIANTS = ("mlp_gauss_tvf", "mlp_disc", "humanoid")


def main():
    if len(sys.argv) < 2:  # driver: one child process per variant, then merge
        import subprocess
        merged, merged_meta = {}, {}
        for tag in VARIANTS:
            subprocess.run([sys.executable, os.path.abspath(__file__), tag], check=True)
            part = os.path.join("/tmp", f"variants_{tag}.npz")
            merged.update(np.load(part))
            merged_meta.update(json.load(open(part + ".json")))
        np.savez_compressed(os.path.join(HERE, "variants_golden.npz"), **merged)
        json.dump(merged_meta, open(os.path.join(HERE, "variants_golden.json"), "w"), indent=1)
        print("wrote", len(merged), "arrays;", sum(v.nbytes for v in merged.values()) / 1e6, "MB raw")
        return
    which = sys.argv[1]
    common = ["--device=cpu", "--env_reward_normalization=off", "--disable_ev=True", "--output_folder=/tmp/ref_golden_out",
              f"--agents={MB}", "--n_steps=4", "--seed=7", "--model_architecture=dual", "--model_encoder=mlp",
              f"--model_hidden_units={256 if which == 'humanoid' else 64}", f"--policy_opt_mini_batch_size={MB}",
              f"--value_opt_mini_batch_size={MB}", f"--distil_opt_mini_batch_size={MB}"]
    if which == "mlp_gauss_tvf":
        run_variant(which, common + ["--env_type=mujoco", "--env_name=Humanoid", "--tvf_enabled=True",
                                     "--tvf_value_heads=8", "--tvf_max_horizon=1000"],
                    (11,), 3, "gaussian", "tanh", True, 64)
    elif which == "humanoid":
        run_variant(which, common + ["--env_type=mujoco", "--env_name=Humanoid", "--tvf_enabled=True",
                                     "--tvf_value_heads=128", "--tvf_max_horizon=30000"],
                    (377,), 17, "gaussian", "tanh", True, 256)
    else:
        run_variant(which, common + ["--env_type=atari", "--tvf_enabled=False"], (4,), 2, "discrete", "relu", False, 64)
    part = os.path.join("/tmp", f"variants_{which}.npz")
    np.savez_compressed(part, **out)
    json.dump(meta, open(part + ".json", "w"), indent=1)
    for k in out:
        if k.endswith("_result"):
            print(k, out[k].tolist())
    print({t: meta[t].get("grad_none") for t in meta})


if __name__ == "__main__":
    main()
