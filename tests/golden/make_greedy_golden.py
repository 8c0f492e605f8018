#!/usr/bin/env python3
"""Generate tests/golden/greedy_golden.npz: greedy actions of the REFERENCE's seeded IMPALA network on 256
observations per network shape (the north star's "greedy actions bit-exact on fixed seeds" gate; model_golden /
shapes_golden hold 8 observations per shape, this is the wide version).

Build container only (needs /root/reference; see ref_shim.py):  python tests/golden/make_greedy_golden.py

Per shape tag (c2 = 4x84x84 / 6 actions, c3 = 3x64x64 / 15 actions, c4 = 4x84x84 / 4 actions) the same seeded
`TVFModel` as make_model_golden.py builds (torch.manual_seed(1): its initial weights are the ones whose sha256
model_golden.json / shapes_golden.json pin).  The observations are NOT stored (7 MB per shape): they are
`np.random.default_rng(OBS_SEED[tag]).integers(0, 256, (256, C, H, W), dtype=uint8)` and the test regenerates them;
their sha256 is stored so a NumPy whose generator differs is detected rather than mis-reported as a parity failure.
Stored (data only): raw_policy [256, n_actions] f32, value [256] f32, the reference's argmax_policy rows reduced to
their index (policy_temperature=0, rl/models.py:475-485), the top-1 / top-2 logit margin of every row.

A freshly initialised policy head (orthogonal x head_scale 0.1) prefers one or two actions on every observation, so a
second pass ("wide_") replaces policy_head.weight by stored seeded normal draws (scale 1) and policy_head.bias by
minus the batch mean of the logits those weights give (stored too): the logits are then centred over the batch and the
greedy choice follows each observation's own feature deviation, spreading over all actions — which is what makes
index-exactness a real test.
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
BATCH = 256
SHAPES = {"c2": ((4, 84, 84), 6), "c3": ((3, 64, 64), 15), "c4": ((4, 84, 84), 4)}
OBS_SEED = {"c2": 1002, "c3": 1003, "c4": 1004}


def main():
    load_reference(["--model_architecture=single", "--model_encoder=impala", "--env_embed_time=False", "--device=cpu",
                    "--env_reward_normalization=off", "--disable_ev=True", "--output_folder=/tmp/ref_golden_out",
                    f"--seed={SEED}"])
    import torch
    from rl import config, models
    args = config.args
    out, meta = {}, {"seed": SEED, "batch": BATCH, "obs_seed": OBS_SEED, "shapes": {k: [list(d), n] for k, (d, n) in SHAPES.items()}}
    for tag, (dims, n_actions) in SHAPES.items():
        torch.manual_seed(SEED)
        model = models.TVFModel(
            encoder="impala", encoder_args=None, input_dims=dims, actions=n_actions, device="cpu",
            architecture="single", dtype=torch.float32, hidden_units=args.model.hidden_units,
            encoder_activation_fn="relu", head_scale=args.model.head_scale, head_bias=args.model.head_bias,
            value_head_names=("ext",))
        x = np.random.default_rng(OBS_SEED[tag]).integers(0, 256, size=(BATCH, *dims), dtype=np.uint8)
        meta.setdefault("obs_sha256", {})[tag] = hashlib.sha256(x.tobytes()).hexdigest()
        meta.setdefault("first_conv_sha256", {})[tag] = hashlib.sha256(
            model.policy_net.encoder.stacks[0].firstconv.weight.detach().numpy().tobytes()).hexdigest()
        with torch.no_grad():
            r0 = model.forward(x, output="policy", policy_temperature=0.0)
            r1 = model.forward(x, output="policy", policy_temperature=1.0)
        raw = r1["raw_policy"].numpy()
        out[f"{tag}_raw_policy"] = raw
        out[f"{tag}_value"] = r1["value"].numpy().reshape(BATCH)
        am = r0["argmax_policy"].numpy()
        assert am.shape == raw.shape and np.all(am.sum(1) == 1)
        out[f"{tag}_greedy_actions"] = am.argmax(1).astype(np.int64)
        assert np.array_equal(out[f"{tag}_greedy_actions"], raw.argmax(1))
        top2 = np.sort(raw, axis=1)[:, -2:]
        out[f"{tag}_logit_margin"] = (top2[:, 1] - top2[:, 0]).astype(np.float32)
        m = out[f"{tag}_logit_margin"]
        print(tag, "margin min", m.min(), "rows below 1e-5:", int((m <= 1e-5).sum()), "actions used:",
              np.bincount(out[f"{tag}_greedy_actions"], minlength=n_actions).tolist())
        head = model.policy_net.policy_head
        hr = np.random.default_rng(OBS_SEED[tag] + 50)
        w = hr.standard_normal(tuple(head.weight.shape)).astype(np.float32)
        with torch.no_grad():
            head.weight.copy_(torch.from_numpy(w))
            head.bias.zero_()
            b = -model.forward(x, output="policy")["raw_policy"].numpy().mean(0).astype(np.float32)
            head.bias.copy_(torch.from_numpy(b))
            r0 = model.forward(x, output="policy", policy_temperature=0.0)
            r1 = model.forward(x, output="policy", policy_temperature=1.0)
        raw = r1["raw_policy"].numpy()
        out[f"{tag}_wide_head_weight"], out[f"{tag}_wide_head_bias"] = w, b
        out[f"{tag}_wide_raw_policy"] = raw
        out[f"{tag}_wide_greedy_actions"] = r0["argmax_policy"].numpy().argmax(1).astype(np.int64)
        assert np.array_equal(out[f"{tag}_wide_greedy_actions"], raw.argmax(1))
        top2 = np.sort(raw, axis=1)[:, -2:]
        m = out[f"{tag}_wide_logit_margin"] = (top2[:, 1] - top2[:, 0]).astype(np.float32)
        print(tag, "wide: margin min", m.min(), "rows below 1e-5:", int((m <= 1e-5).sum()), "actions used:",
              np.bincount(out[f"{tag}_wide_greedy_actions"], minlength=n_actions).tolist())
    np.savez_compressed(os.path.join(HERE, "greedy_golden.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "greedy_golden.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
