#!/usr/bin/env python3
"""Generate tests/golden/tvf_mask_golden.npz: the REFERENCE's TVF feature masks (rl/models.py:386-427,
`--tvf_feature_sparsity` / `--tvf_feature_window`) on CPU: the masked + rescaled initial TVF head, the 0 / 1 mask,
forward outputs, and what one optimiser step does to the head (masked weights become non-zero in the step and are
zeroed again by `mask_feature_weights` at the next forward that evaluates the head, rl/models.py:494-497).

Build container only (needs /root/reference; see ref_shim.py):  python tests/golden/make_tvf_mask_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_shim import load_reference  # noqa: E402

MB, HIDDEN, DIMS, NA = 16, 64, (11,), 3
CASES = {"sparsity": dict(tvf_feature_sparsity=0.5, tvf_feature_window=-1),
         "window": dict(tvf_feature_sparsity=0.0, tvf_feature_window=16)}


def main():
    rl = load_reference(["--device=cpu", "--env_reward_normalization=off", "--disable_ev=True",
                         "--output_folder=/tmp/ref_golden_out", f"--agents={MB}", "--n_steps=4", "--seed=7",
                         "--model_architecture=dual", "--model_encoder=mlp", f"--model_hidden_units={HIDDEN}",
                         "--env_type=mujoco", "--env_name=Humanoid", "--tvf_enabled=True", "--tvf_value_heads=8",
                         "--tvf_max_horizon=1000"])
    import torch
    from rl import config, models
    import rl.tvf
    args = config.args
    out, meta = {}, {}
    horizons, weights = rl.tvf.get_value_head_horizons(args.tvf.value_heads, args.tvf.max_horizon,
                                                       args.tvf.head_spacing, include_weight=True)
    out["horizons"] = np.asarray(horizons)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((MB, *DIMS)).astype(np.float32)
    target = rng.standard_normal((MB, len(horizons))).astype(np.float32)
    out["x"], out["target"] = x, target
    for tag, kw in CASES.items():
        torch.manual_seed(7)
        model = models.TVFModel(
            encoder="mlp", encoder_args=None, input_dims=DIMS, actions=NA, device="cpu", architecture="dual",
            dtype=torch.float32, hidden_units=HIDDEN, encoder_activation_fn="tanh", tvf_fixed_head_horizons=horizons,
            tvf_fixed_head_weights=weights, head_scale=args.model.head_scale, head_bias=args.model.head_bias,
            value_head_names=("ext",), **kw)
        net = model.value_net
        m = meta.setdefault(tag, dict(kw, hidden=HIDDEN, input_dims=list(DIMS), n_actions=NA,
                                      head_scale=args.model.head_scale, head_bias=args.model.head_bias, params={}))
        for prefix, n_ in (("policy_net", model.policy_net), ("value_net", model.value_net)):
            for n, p in n_.named_parameters():
                a = p.detach().numpy()
                m["params"][f"{prefix}.{n}"] = {"shape": list(a.shape), "sha256": hashlib.sha256(a.tobytes()).hexdigest()}
                out[f"{tag}_init_{prefix}.{n}"] = a.copy()
        out[f"{tag}_mask"] = net.tvf_features_mask.numpy().copy()
        out[f"{tag}_w0"] = net.tvf_head.weight.detach().numpy().copy()
        with torch.no_grad():
            out[f"{tag}_fwd0_tvf_value"] = model.forward(x, output="value")["tvf_value"].numpy().copy()
        # one plain optimiser step on a squared error of the TVF heads (lr large enough to move masked weights visibly)
        opt = torch.optim.Adam(net.parameters(), lr=1e-2, eps=1e-5)
        opt.zero_grad(set_to_none=True)
        pred = model.forward(x, output="value")["tvf_value"][..., 0]
        loss = (0.5 * (pred - torch.from_numpy(target)) ** 2).mean()
        loss.backward()
        for n, p in net.named_parameters():
            if p.grad is not None:
                out[f"{tag}_grad_{n}"] = p.grad.detach().numpy().copy()
        opt.step()
        raw = net.tvf_head.weight.detach().numpy().copy()  # masked entries moved by the step
        m["masked_entries_nonzero_after_step"] = int(np.count_nonzero(raw[out[f"{tag}_mask"] == 0]))
        with torch.no_grad():
            out[f"{tag}_fwd1_tvf_value"] = model.forward(x, output="value")["tvf_value"].numpy().copy()
        out[f"{tag}_w1"] = net.tvf_head.weight.detach().numpy().copy()  # after mask_feature_weights
        for n, p in net.named_parameters():
            out[f"{tag}_after_{n}"] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "tvf_mask_golden.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "tvf_mask_golden.json"), "w"), indent=1)
    print("wrote", len(out), "arrays;", {t: (int(out[f"{t}_mask"].sum()), meta[t]["masked_entries_nonzero_after_step"]) for t in CASES})


if __name__ == "__main__":
    main()
