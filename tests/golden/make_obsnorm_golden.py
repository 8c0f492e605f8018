#!/usr/bin/env python3
"""Generate tests/golden/obsnorm_golden.npz by running the REFERENCE's observation normalisation on CPU.

Build container only (needs /root/reference; see ref_shim.py):  python tests/golden/make_obsnorm_golden.py

What is captured (data only), for two models built with observation_normalization=True:
  mlp   input_dims (11,), float32 observations with per-feature offsets and scales (mujoco-like);
  img   input_dims (4, 84, 84) is too big for a fixture, so the uint8 case uses (2, 12, 12) with the mlp
        encoder (the normalisation code never looks at the encoder).
For each: three batches; after each `perform_normalization(x, update_normalization=True)` (rl/models.py:666-694)
the running mean / var / count (obs_rms), the float32 constants _mu / _std and the normalised batch; then the
normalised output of a fourth batch WITHOUT an update.  For the mlp model also a seeded model's forward
(log_policy, value) on that fourth batch, which goes through normalisation inside TVFModel.forward (:783-784).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_shim import load_reference  # noqa: E402

SEED = 5


def main():
    load_reference(["--model_architecture=single", "--model_encoder=mlp", "--env_embed_time=False", "--device=cpu",
                    "--output_folder=/tmp/ref_golden_out", "--agents=8", "--n_steps=4", f"--seed={SEED}",
                    "--observation_normalization=True"])
    import torch
    from rl import config, models
    args = config.args
    out = {}
    rng = np.random.default_rng(SEED)
    cases = {
        "mlp": ((11,), lambda b: (rng.normal(size=(b, 11)) * np.linspace(0.01, 30, 11) + np.linspace(-5, 50, 11)).astype(np.float32)),
        "img": ((2, 12, 12), lambda b: rng.integers(0, 256, size=(b, 2, 12, 12), dtype=np.uint8)),
    }
    for tag, (dims, gen) in cases.items():
        torch.manual_seed(SEED)
        model = models.TVFModel(
            encoder="mlp", encoder_args=None, input_dims=dims, actions=3, device="cpu", architecture="single",
            dtype=torch.float32, hidden_units=64, encoder_activation_fn="tanh", observation_normalization=True,
            head_scale=args.model.head_scale, head_bias=args.model.head_bias, value_head_names=("ext",))
        out[f"{tag}_norm_eps"] = np.float32(model.norm_eps)
        out[f"{tag}_count_init"] = np.float64(model.obs_rms.count)
        for i, b in enumerate((16, 7, 32)):
            x = gen(b)
            out[f"{tag}_x{i}"] = x
            y = model.perform_normalization(model.prep_for_model(x), update_normalization=True)
            out[f"{tag}_y{i}"] = y.numpy()
            out[f"{tag}_mean{i}"] = np.asarray(model.obs_rms.mean, np.float64)
            out[f"{tag}_var{i}"] = np.asarray(model.obs_rms.var, np.float64)
            out[f"{tag}_count{i}"] = np.float64(model.obs_rms.count)
            out[f"{tag}_mu{i}"] = model._mu.numpy()
            out[f"{tag}_std{i}"] = model._std.numpy()
        x = gen(9)
        out[f"{tag}_x3"] = x
        out[f"{tag}_y3"] = model.perform_normalization(model.prep_for_model(x)).numpy()
        if tag == "mlp":
            with torch.no_grad():
                r = model.forward(x, output="policy")
            out["mlp_fwd_log_policy"] = r["log_policy"].numpy()
            out["mlp_fwd_value"] = r["value"].numpy()
            out["mlp_head_scale"] = np.float32(args.model.head_scale)
            out["mlp_head_bias"] = np.bool_(args.model.head_bias)
    np.savez_compressed(os.path.join(HERE, "obsnorm_golden.npz"), **out)
    print("wrote obsnorm_golden.npz:", {k: getattr(v, "shape", ()) for k, v in out.items() if k.startswith("mlp_m") or "fwd" in k})


if __name__ == "__main__":
    main()
