"""CPU: oracle/model_torch.py (the plain-PyTorch restatement of the network + PPO loss) against the
reference's own outputs in tests/golden/model_golden.npz.  Same torch CPU operators as the reference,
so agreement is at fp32 round-off; this pins the oracle used by the GPU parity tests and by
bench.py's cpu_baseline."""
import json
import os

import numpy as np
import torch

from oracle import model_torch as R
from ppo_amd.models import ImpalaSpec, init_impala_parameters


def _setup(golden_dir):
    g = np.load(os.path.join(golden_dir, "model_golden.npz"))
    meta = json.load(open(os.path.join(golden_dir, "model_golden.json")))
    torch.manual_seed(meta["seed"])
    init = init_impala_parameters(ImpalaSpec(tuple(meta["input_dims"]), hidden_units=meta["hidden_units"]),
                                  meta["n_actions"], 1, meta["head_scale"], meta["head_bias"])
    return g, meta, init


def test_forward_matches_reference(golden_dir):
    g, meta, init = _setup(golden_dir)
    with torch.no_grad():
        out = R.forward(init, torch.from_numpy(g["fwd_x"]).float() / 255.0)
    for k in ("raw_policy", "log_policy", "value", "advantage"):
        ref = g[f"fwd_{k}"]
        assert np.abs(out[k].numpy() - ref).max() <= 1e-6 * max(np.abs(ref).max(), 1.0), k
    assert np.array_equal(out["raw_policy"].argmax(1).numpy(), g["fwd_greedy_actions"])


def test_loss_and_gradients_match_reference(golden_dir):
    g, meta, init = _setup(golden_dir)
    sd = {k: v.clone().requires_grad_(True) for k, v in init.items()}
    out = R.forward(sd, torch.from_numpy(g["mb0_prev_state"]).float() / 255.0)
    loss = R.ppo_loss(out, torch.from_numpy(g["mb0_actions"]), torch.from_numpy(g["mb0_log_pac"]),
                      torch.from_numpy(g["mb0_advantages"]), torch.from_numpy(g["mb0_returns"]),
                      eps=meta["ppo_epsilon"], ent_coef=meta["entropy_bonus"], vf_coef=meta["ppo_vf_coef"])
    loss.backward()
    assert abs(loss.item() - g["mb0_result"][0]) < 1e-6
    stride = meta["dense_row_stride"]
    for name in meta["param_names"]:
        if meta["params"][name].get("grad_none"):
            assert sd[name].grad is None
            continue
        got = sd[name].grad.numpy()
        got = got[::stride] if name == "encoder.dense.weight" else got
        ref = g["grad0_" + name]
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max(), name


def test_mlp_forward_matches_reference(golden_dir):
    """oracle/model_torch.mlp_forward (bench.py's cpu_baseline for the MLP + TVF config) against the reference's own
    forward outputs for the MLP variants (tests/golden/variants_golden.npz): both nets of the dual architecture."""
    from ppo_amd.models import MLPSpec, init_parameters
    g = np.load(os.path.join(golden_dir, "variants_golden.npz"))
    meta = json.load(open(os.path.join(golden_dir, "variants_golden.json")))
    for tag in ("mlp_gauss_tvf", "mlp_disc"):
        m = meta[tag]
        K = len(g[f"{tag}_tvf_horizons"]) if f"{tag}_tvf_horizons" in g else 0
        torch.manual_seed(7)
        spec = MLPSpec(tuple(m["input_dims"]), hidden_units=m["hidden"])
        pol = init_parameters(spec, m["n_actions"], 1, m["head_scale"], m["head_bias"], K)
        val = init_parameters(spec, m["n_actions"], 1, m["head_scale"], m["head_bias"], K)
        x = torch.from_numpy(g[f"{tag}_x"])
        with torch.no_grad():
            op, ov = R.mlp_forward(pol, x, m["activation"]), R.mlp_forward(val, x, m["activation"])
        for key, got in (("policy_raw_policy", op["raw_policy"]), ("value_value", ov["value"])):
            ref = g[f"{tag}_fwd_full_{key}"]
            assert np.abs(got.numpy().reshape(ref.shape) - ref).max() <= 1e-6 * max(np.abs(ref).max(), 1.0), (tag, key)
        if K:
            ref = g[f"{tag}_fwd_value_tvf_value"]
            assert np.abs(ov["tvf_value"].numpy().reshape(ref.shape) - ref).max() <= 1e-6 * max(np.abs(ref).max(), 1.0)
