"""CPU: our parameter initialiser reproduces the reference's initial weights bit for bit.

model_golden.json holds sha256 / moments of every parameter of a reference
TVFModel(impala, single) built under torch.manual_seed(1) (tests/golden/make_model_golden.py);
ppo_amd.models.init_impala_parameters must draw the same values from the same seed
(reference construction order: rl/models.py:348-368, :73-84, rl/impala.py:60-62,96-100;
initialisers: rl/tensor_utilities.py:40-95)."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from ppo_amd.models import ImpalaSpec, init_impala_parameters


def _metas(golden_dir):
    shapes = json.load(open(os.path.join(golden_dir, "shapes_golden.json")))
    return {"c2": json.load(open(os.path.join(golden_dir, "model_golden.json"))), "c3": shapes["c3"], "c4": shapes["c4"]}


@pytest.mark.parametrize("tag,n_params", [("c2", 1092579), ("c3", 630126), ("c4", 1091549)])
def test_initial_parameters_match_reference_bitwise(golden_dir, tag, n_params):
    """c2: 4x84x84 / 6 actions (Pong), c3: 3x64x64 / 15 actions (procgen), c4: 4x84x84 / 4 actions (Breakout)."""
    meta = _metas(golden_dir)[tag]
    torch.manual_seed(meta["seed"])
    spec = ImpalaSpec(tuple(meta["input_dims"]), hidden_units=meta["hidden_units"])
    init = init_impala_parameters(spec, meta["n_actions"], 1, meta["head_scale"], meta["head_bias"])
    assert set(init) == set(meta["params"])
    total = 0
    for name, info in meta["params"].items():
        a = init[name].numpy()
        assert list(a.shape) == info["shape"], name
        assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() == info["sha256"], name
        total += a.size
    assert total == n_params  # c2: SURVEY.md §8a R9 [probed] parameter count at 6 actions


def test_geometry():
    spec = ImpalaSpec((4, 84, 84))
    assert spec.out_shape == (32, 11, 11) and spec.flat == 3872
    assert ImpalaSpec((3, 64, 64)).flat == 32 * 8 * 8


def test_mlp_dual_tvf_initial_parameters_match_reference_bitwise(golden_dir):
    """MLP encoder, dual architecture (policy_net then value_net drawn in sequence), with and without the
    TVF head (tests/golden/make_variants_golden.py; rl/models.py:157-161, 364-384, 605-607)."""
    from ppo_amd.models import MLPSpec, init_parameters
    meta = json.load(open(os.path.join(golden_dir, "variants_golden.json")))
    gold = np.load(os.path.join(golden_dir, "variants_golden.npz"))
    for tag in ("mlp_gauss_tvf", "mlp_disc"):
        m = meta[tag]
        n_tvf = len(gold[f"{tag}_tvf_horizons"]) if f"{tag}_tvf_horizons" in gold else 0
        torch.manual_seed(7)
        spec = MLPSpec(tuple(m["input_dims"]), hidden_units=m["hidden"])
        for prefix in ("policy_net", "value_net"):
            init = init_parameters(spec, m["n_actions"], 1, m["head_scale"], m["head_bias"], n_tvf)
            want = {k[len(prefix) + 1:]: v for k, v in m["params"].items() if k.startswith(prefix + ".")}
            assert set(init) == set(want), (tag, prefix)
            for name, info in want.items():
                a = np.ascontiguousarray(init[name].numpy())
                assert list(a.shape) == info["shape"], name
                assert hashlib.sha256(a.tobytes()).hexdigest() == info["sha256"], (tag, prefix, name)


@pytest.mark.parametrize("tag", ["sparsity", "window"])
def test_tvf_feature_masks_and_masked_initial_head_match_reference_bitwise(golden_dir, tag):
    """--tvf_feature_sparsity / --tvf_feature_window (rl/models.py:386-421; tests/golden/make_tvf_mask_golden.py): the
    0 / 1 mask and the masked, rescaled initial TVF head from the same seeds as the reference, bit for bit (on this
    host: the fixture's own)."""
    from ppo_amd.models import MLPSpec, init_parameters, tvf_feature_mask
    meta = json.load(open(os.path.join(golden_dir, "tvf_mask_golden.json")))[tag]
    gold = np.load(os.path.join(golden_dir, "tvf_mask_golden.npz"))
    K, H = len(gold["horizons"]), meta["hidden"]
    scaled = tvf_feature_mask(K, H, meta["tvf_feature_sparsity"], meta["tvf_feature_window"])
    assert np.array_equal(torch.gt(scaled, 0).to(torch.uint8).numpy(), gold[f"{tag}_mask"])
    torch.manual_seed(7)
    spec = MLPSpec(tuple(meta["input_dims"]), hidden_units=H)
    for prefix in ("policy_net", "value_net"):
        init = init_parameters(spec, meta["n_actions"], 1, meta["head_scale"], meta["head_bias"], K)
        init["tvf_head.weight"] = init["tvf_head.weight"] * scaled  # DualHeadNet._build_tvf_feature_mask
        for name, t in init.items():
            assert np.array_equal(t.numpy(), gold[f"{tag}_init_{prefix}.{name}"]), (prefix, name)
    assert np.array_equal(gold[f"{tag}_init_value_net.tvf_head.weight"], gold[f"{tag}_w0"])
