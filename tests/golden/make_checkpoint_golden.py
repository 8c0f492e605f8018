#!/usr/bin/env python3
"""Generate tests/golden/checkpoint_golden.json: the KEY TREE of what the REFERENCE's Runner.save_checkpoint hands to
torch.save (rl/rollout.py:394-453) — keys, container types, tensor dtypes / shapes, nothing else — for two set-ups:

  impala_single   IMPALA 4x84x84 / 6 actions / single architecture (the benchmark configuration)
  mlp_dual        mlp / dual / discrete (2 actions): policy, value and distil optimisers

Build container only (needs /root/reference; see ref_shim.py):  python tests/golden/make_checkpoint_golden.py

One policy (+ value + distil) minibatch and optimiser step is run first so that the optimisers hold state.
torch.save is intercepted (the dict is described, never written), logs and env state are disabled (the reference
pickles its Logger object and gym wrappers there: not data).  torch.optim's `param_groups` keys are those of the
torch in this container (2.10), a superset of the reference's pinned 1.12.1 — the test compares the core keys only.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_shim import load_reference  # noqa: E402

MB = 8


def describe(v, depth=0):
    import torch
    if isinstance(v, torch.Tensor):
        return {"__tensor__": str(v.dtype).replace("torch.", ""), "shape": list(v.shape)}
    if isinstance(v, np.ndarray):
        return {"__ndarray__": str(v.dtype), "shape": list(v.shape)}
    if isinstance(v, dict):
        return {"__dict__": {str(k): describe(x, depth + 1) for k, x in v.items()},
                "key_type": sorted({type(k).__name__ for k in v})}
    if isinstance(v, (list, tuple)):
        kinds = [describe(x, depth + 1) for x in v]
        same = all(k == kinds[0] for k in kinds) if kinds else True
        return {"__seq__": type(v).__name__, "len": len(v), "items": kinds[:1] if same else kinds}
    if v is None or isinstance(v, (bool, int, float, str)):
        return {"__scalar__": type(v).__name__}
    if isinstance(v, np.generic):
        return {"__scalar__": "numpy." + type(v).__name__}
    return {"__object__": type(v).__module__ + "." + type(v).__name__}


def run(tag):
    common = ["--device=cpu", "--env_reward_normalization=off", "--disable_ev=True", "--output_folder=/tmp/ref_golden_out",
              f"--agents={MB}", "--n_steps=4", "--seed=3", f"--policy_opt_mini_batch_size={MB}",
              f"--value_opt_mini_batch_size={MB}", f"--distil_opt_mini_batch_size={MB}", "--env_embed_time=False",
              "--checkpoint_compression=False"]
    if tag == "impala_single":
        flags, dims, nA, enc, arch = common + ["--model_architecture=single", "--model_encoder=impala"], (4, 84, 84), 6, "impala", "single"
    else:
        flags, dims, nA, enc, arch = common + ["--model_architecture=dual", "--model_encoder=mlp", "--model_hidden_units=64",
                                               "--tvf_enabled=False"], (4,), 2, "mlp", "dual"
    load_reference(flags)
    import torch
    from rl import config, logger, models, rollout
    args = config.args
    torch.manual_seed(3)
    model = models.TVFModel(encoder=enc, encoder_args=None, input_dims=dims, actions=nA, device="cpu", architecture=arch,
                            dtype=torch.float32, hidden_units=args.model.hidden_units, encoder_activation_fn="relu",
                            head_scale=args.model.head_scale, head_bias=args.model.head_bias, value_head_names=("ext",))
    runner = rollout.Runner(model, logger.Logger(), action_dist="discrete")
    rng = np.random.default_rng(3)
    if enc == "impala":
        x = torch.from_numpy(rng.integers(0, 256, size=(MB, *dims), dtype=np.uint8))
    else:
        x = torch.from_numpy(rng.standard_normal((MB, *dims)).astype(np.float32))
    with torch.no_grad():
        cur = model.forward(x, output="policy")
    lp = torch.log_softmax(cur["raw_policy"], dim=1)
    actions = torch.from_numpy(rng.integers(0, nA, size=(MB,)).astype(np.int64))
    data = {"prev_state": x, "actions": actions, "log_policy": lp, "log_pac": lp[range(MB), actions],
            "advantages": torch.from_numpy(rng.normal(size=(MB,)).astype(np.float32)),
            "returns": torch.from_numpy(rng.normal(size=(MB, 1)).astype(np.float32))}
    runner.policy_optimizer.zero_grad(set_to_none=True)
    runner.train_policy_minibatch(data, loss_scale=1.0)
    runner.optimizer_step(runner.policy_optimizer, "policy")
    if arch == "dual":
        runner.value_optimizer.zero_grad(set_to_none=True)
        runner.train_value_minibatch({"prev_state": x, "returns": data["returns"]}, loss_scale=1.0)
        runner.optimizer_step(runner.value_optimizer, "value")
        if runner.distil_optimizer is not None:
            runner.distil_optimizer.zero_grad(set_to_none=True)
            for p in model.policy_net.parameters():
                p.grad = None
            runner.train_distil_minibatch({"prev_state": x, "distil_targets": data["returns"][:, 0], "old_log_policy": lp,
                                           "old_raw_policy": cur["raw_policy"]}, loss_scale=1.0)
            runner.optimizer_step(runner.distil_optimizer, "distil")
    captured = {}
    real_save = torch.save
    torch.save = lambda obj, f, **kw: captured.update(obj)
    try:
        runner.save_checkpoint(f"/tmp/ref_golden_out_ckpt_{tag}.pt", 12345, disable_log=True, disable_env_state=True)
    finally:
        torch.save = real_save
    tree = describe(captured)
    groups = {}
    for k, v in captured.items():
        if k.endswith("optimizer_state_dict"):
            g = v["param_groups"][0]
            groups[k] = {kk: (list(vv) if isinstance(vv, tuple) else vv) for kk, vv in g.items() if kk != "params"}
            groups[k]["n_params"] = len(g["params"])
            groups[k]["state_indices"] = sorted(int(i) for i in v["state"].keys())
    return {"tree": tree, "param_groups": groups,
            "policy_param_names": [n for n, _ in model.policy_net.named_parameters()],
            "torch_version": torch.__version__}


def main():
    if len(sys.argv) > 1:
        json.dump(run(sys.argv[1]), open(f"/tmp/ckpt_tree_{sys.argv[1]}.json", "w"))
        return
    import subprocess
    out = {}
    for tag in ("impala_single", "mlp_dual"):
        subprocess.run([sys.executable, os.path.abspath(__file__), tag], check=True)
        out[tag] = json.load(open(f"/tmp/ckpt_tree_{tag}.json"))
    json.dump(out, open(os.path.join(HERE, "checkpoint_golden.json"), "w"), indent=1, sort_keys=True)
    for tag, v in out.items():
        print(tag, sorted(v["tree"]["__dict__"].keys()), v["param_groups"])


if __name__ == "__main__":
    main()
