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
  G5k the reference's own kink decisions during that first backward's forward pass: the sign of every
      ReLU input (bit-packed) and the tap each max-pool window selected (0..8 = dy*3+dx inside the 3x3
      window), captured with forward hooks.  With these a test can COUNT the elements on which another
      fp32 implementation decides differently and compare gradients with the decisions aligned.
  shapes_golden.npz: the same G4 + one G5 step (+ G5k) for BASELINE configs 3 and 4's network shapes:
      procgen 3x64x64 / 15 actions ("c3_") and atari 4x84x84 / 4 actions ("c4_").
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
FWD_BATCH = 8
MB = 16
DENSE_ROW_STRIDE = 16


def sub(name, t):
    """Full tensor, except the big dense matrix which is sampled by rows."""
    a = t.detach().cpu().numpy()
    if name.endswith("encoder.dense.weight"):
        return a[::DENSE_ROW_STRIDE].copy()
    return a.copy()


class KinkRecorder:
    """Forward hooks on the reference's policy_net that record which side of every kink its forward pass took:
    `relu_<module>` = bit-packed (input > 0) of every module that consumes a ReLU output (block convolutions, the
    dense layer, the heads: rl/impala.py:73-78, rl/models.py:97, 467), `pool_<si>` = the window tap (0..8) each
    max-pool output took (rl/impala.py:105)."""

    def __init__(self, torch, net):
        self.torch, self.on, self.out, self.handles = torch, False, {}, []
        enc = net.encoder
        for si, stack in enumerate(enc.stacks):
            self.handles.append(stack.firstconv.register_forward_hook(self._pool(si)))
            for bi, block in enumerate(stack.blocks):
                for cname in ("conv0", "conv1"):
                    self.handles.append(getattr(block, cname).register_forward_pre_hook(
                        self._relu(f"encoder.stacks.{si}.blocks.{bi}.{cname}")))
        self.handles.append(enc.dense.register_forward_pre_hook(self._relu("encoder.dense")))
        self.handles.append(net.policy_head.register_forward_pre_hook(self._relu("heads")))

    def _relu(self, name):
        def hook(_m, inp):
            if self.on:
                self.out["relu_" + name] = np.packbits((inp[0].detach() > 0).numpy().reshape(-1))
        return hook

    def _pool(self, si):
        def hook(_m, _inp, outp):
            if self.on:
                F = self.torch.nn.functional
                c = outp.detach()
                _, flat = F.max_pool2d(c, kernel_size=3, stride=2, padding=1, return_indices=True)
                H, W = c.shape[2], c.shape[3]
                Ho, Wo = flat.shape[2], flat.shape[3]
                iy, ix = flat // W, flat % W
                oy = self.torch.arange(Ho).view(1, 1, Ho, 1)
                ox = self.torch.arange(Wo).view(1, 1, 1, Wo)
                tap = (iy - (2 * oy - 1)) * 3 + (ix - (2 * ox - 1))
                assert int(tap.min()) >= 0 and int(tap.max()) <= 8
                self.out[f"pool_{si}"] = tap.numpy().astype(np.uint8)
        return hook

    def remove(self):
        for h in self.handles:
            h.remove()


def capture(input_dims, n_actions, steps, after_steps, env_flags=()):
    """One seeded reference TVFModel of the given shape: G4 forward, `steps` G5 optimiser steps, G5k kinks of step 0."""
    rl = load_reference([
        "--model_architecture=single", "--model_encoder=impala", "--env_embed_time=False", "--device=cpu",
        "--env_reward_normalization=off", "--disable_ev=True", "--output_folder=/tmp/ref_golden_out",
        f"--agents={MB}", "--n_steps=4", f"--seed={SEED}", f"--policy_opt_mini_batch_size={MB}", *env_flags])
    import torch
    from rl import config, logger, models, rollout
    args = config.args
    torch.manual_seed(SEED)
    model = models.TVFModel(
        encoder="impala", encoder_args=None, input_dims=input_dims, actions=n_actions, device="cpu",
        architecture="single", dtype=torch.float32, hidden_units=args.model.hidden_units,
        encoder_activation_fn="relu", head_scale=args.model.head_scale, head_bias=args.model.head_bias,
        value_head_names=("ext",))
    out = {}
    meta = {"seed": SEED, "n_actions": n_actions, "input_dims": input_dims, "hidden_units": args.model.hidden_units,
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
    x = rng.integers(0, 256, size=(FWD_BATCH, *input_dims), dtype=np.uint8)
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
    kinks = KinkRecorder(torch, model.policy_net)
    for step in range(steps):
        xs = rng.integers(0, 256, size=(MB, *input_dims), dtype=np.uint8)
        with torch.no_grad():
            cur = model.forward(xs, output="policy")
        # behaviour policy = current policy perturbed, so ratios straddle the clip range
        old_logits = cur["raw_policy"] + 0.5 * torch.from_numpy(rng.normal(size=(MB, n_actions)).astype(np.float32))
        old_lp = torch.log_softmax(old_logits, dim=1)
        actions = torch.from_numpy(rng.integers(0, n_actions, size=(MB,)).astype(np.int64))
        log_pac = old_lp[range(MB), actions]
        adv = torch.from_numpy(rng.normal(size=(MB,)).astype(np.float32))
        ret = torch.from_numpy(rng.normal(size=(MB, 1)).astype(np.float32))
        data = {"prev_state": torch.from_numpy(xs), "actions": actions, "log_policy": old_lp, "log_pac": log_pac,
                "advantages": adv, "returns": ret}
        for k, v in data.items():
            out[f"mb{step}_{k}"] = v.numpy()
        opt.zero_grad(set_to_none=True)
        kinks.on = step == 0
        res = runner.train_policy_minibatch(data, loss_scale=1.0)
        kinks.on = False
        out[f"mb{step}_result"] = np.asarray([res["loss"], res["kl_approx"], res["kl_true"], res["clip_frac"]], np.float64)
        if step == 0:
            for n, p in model.policy_net.named_parameters():
                if p.grad is not None:
                    out["grad0_" + n] = sub(n, p.grad)
                else:
                    meta["params"][n]["grad_none"] = True
            for k, v in kinks.out.items():
                out["kink0_" + k] = v
        gn = runner.optimizer_step(opt, "policy")
        out[f"mb{step}_grad_norm"] = np.asarray(gn, np.float64)
        if step + 1 in after_steps:
            for n, p in model.policy_net.named_parameters():
                out[f"param_after{step + 1}_" + n] = sub(n, p)
    kinks.remove()
    meta["param_names"] = names
    return out, meta


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "model"):
        out, meta = capture((4, 84, 84), 6, steps=4, after_steps=(1, 4))
        np.savez_compressed(os.path.join(HERE, "model_golden.npz"), **out)
        json.dump(meta, open(os.path.join(HERE, "model_golden.json"), "w"), indent=1)
        print("model_golden: wrote", len(out), "arrays;", sum(v.nbytes for v in out.values()) / 1e6, "MB raw")
        print({k: out[k].tolist() for k in out if k.endswith("_result") or k.endswith("grad_norm")})
        print("margins", out["fwd_logit_margin"])
    if which in ("all", "shapes"):
        # the reference is one process-global config + module set: each shape is captured in a fresh interpreter
        import subprocess
        if len(sys.argv) > 2:
            tag = sys.argv[2]
            dims, nA = {"c3": ((3, 64, 64), 15), "c4": ((4, 84, 84), 4)}[tag]
            out, meta = capture(dims, nA, steps=1, after_steps=(1,))
            np.savez_compressed(f"/tmp/shapes_{tag}.npz", **out)
            json.dump(meta, open(f"/tmp/shapes_{tag}.json", "w"))
            return
        allout, allmeta = {}, {}
        for tag in ("c3", "c4"):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "shapes", tag])
            z = np.load(f"/tmp/shapes_{tag}.npz")
            for k in z.files:
                allout[f"{tag}_{k}"] = z[k]
            allmeta[tag] = json.load(open(f"/tmp/shapes_{tag}.json"))
            print(tag, "margins", z["fwd_logit_margin"], "result", z["mb0_result"], "grad_norm", z["mb0_grad_norm"])
        np.savez_compressed(os.path.join(HERE, "shapes_golden.npz"), **allout)
        json.dump(allmeta, open(os.path.join(HERE, "shapes_golden.json"), "w"), indent=1)
        print("shapes_golden: wrote", len(allout), "arrays;", sum(v.nbytes for v in allout.values()) / 1e6, "MB raw")


if __name__ == "__main__":
    main()
