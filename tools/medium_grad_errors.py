"""Per-parameter gradient difference of --precision=medium against the exact path on the model fixture's minibatch,
with the switches that select which layers run as split-bf16 products (what tests/test_bf16x3_gpu.py bounds by 2e-3).
usage (GPU box): python tools/medium_grad_errors.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_amd import models  # noqa: E402

here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
g = np.load(os.path.join(here, "model_golden.npz"))
meta = json.load(open(os.path.join(here, "model_golden.json")))


def build(precision):
    torch.manual_seed(meta["seed"])
    return models.DualHeadNet("impala", tuple(meta["input_dims"]), meta["n_actions"], hidden_units=meta["hidden_units"],
                              head_scale=meta["head_scale"], head_bias=meta["head_bias"], device="cuda", precision=precision)


x = torch.from_numpy(g["fwd_x"]).cuda()
B = x.shape[0]
rng = np.random.default_rng(0)
actions = torch.from_numpy(rng.integers(0, meta["n_actions"], B).astype(np.int32)).cuda()
adv = torch.from_numpy(rng.normal(size=B).astype(np.float32)).cuda()
ret = torch.from_numpy(rng.normal(size=(B, 1)).astype(np.float32)).cuda()
hi = build("high")
o_hi = hi.forward(x)
old_lp = o_hi["log_policy"].clone()
old_pac = (old_lp.gather(1, actions.long()[:, None])[:, 0] - 0.1).contiguous()


def grads_of(net):
    net.grad.zero_()
    net.ppo_minibatch(x, actions, old_pac, old_lp, adv, ret)
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in net.grads.items()}


gh = grads_of(hi)
print("batch", B)
for conv, wgrad in ((0, 0), (0, 1), (1, 0), (1, 1)):
    models.SPLIT_CONV, models.SPLIT_WGRAD = conv, wgrad
    md = build("medium")
    gm = grads_of(md)
    rows = []
    for k, a in gh.items():
        s = float(a.abs().max())
        if s > 0:
            rows.append((float((gm[k] - a).abs().max()) / s, float((gm[k] - a).norm()) / float(a.norm()), k))
    rows.sort(reverse=True)
    print(f"SPLIT_CONV={conv} SPLIT_WGRAD={wgrad}: worst max-rel {rows[0][0]:.2e} ({rows[0][2]}), worst l2-rel {max(r[1] for r in rows):.2e}")
    for r in rows[:4]:
        print(f"     {r[2]:48s} max-rel {r[0]:.2e}  l2-rel {r[1]:.2e}")
