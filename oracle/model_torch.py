"""oracle/model_torch.py — TEST INFRASTRUCTURE, NOT PRODUCT.

Plain-PyTorch (CPU or any device, any float dtype) restatement of the reference's
single-architecture IMPALA policy/value network and PPO loss: the "plain PyTorch reference of the
same op" for the floating-point kernels, and — because the reference's own CPU path IS these torch
operators (`--device=cpu`, rl/config.py:731) — the `cpu_baseline` leg of bench.py.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline may import it; ppo_amd never does.

Follows rl/models.py:90-99,433-508 and rl/impala.py:69-82,102-109 of the reference; takes a
``state_dict`` with the reference's key names (relative to policy_net).

Parity: PINNED.  tests/golden/model_golden.npz (outputs of the imported reference) is reproduced by
`forward` / `ppo_loss` to fp32 round-off (loss 0.98257291 on minibatch 0, identical digits; all
gradients within 2e-6 of the reference's), checked in tests/test_oracle_model.py.
"""
import torch
import torch.nn.functional as F


def forward(sd, x, n_stacks=3, n_block=2):
    """x: [B,C,H,W] float (already scaled).  Returns dict(raw_policy, log_policy, value, advantage, h)."""
    for si in range(n_stacks):
        p = f"encoder.stacks.{si}."
        x = F.conv2d(x, sd[p + "firstconv.weight"], sd[p + "firstconv.bias"], padding=1)
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        for bi in range(n_block):
            b = p + f"blocks.{bi}."
            r = F.conv2d(F.relu(x), sd[b + "conv0.weight"], sd[b + "conv0.bias"], padding=1)
            r = F.conv2d(F.relu(r), sd[b + "conv1.weight"], sd[b + "conv1.bias"], padding=1)
            x = x + r
    flat = x.reshape(x.shape[0], -1)
    h = F.linear(F.relu(flat), sd["encoder.dense.weight"], sd["encoder.dense.bias"])
    f = F.relu(h)
    raw = F.linear(f, sd["policy_head.weight"], sd.get("policy_head.bias"))
    return {"raw_policy": raw, "log_policy": F.log_softmax(raw, dim=1),
            "value": F.linear(f, sd["value_head.weight"], sd.get("value_head.bias")),
            "advantage": F.linear(f, sd["advantage_head.weight"], sd.get("advantage_head.bias")), "h": h}


def mlp_forward(sd, x, activation="tanh"):
    """The MLP encoder nets (StandardMLP rl/models.py:148-169: fc1 -> tanh -> fc2; DualHeadNet's encoder activation and
    heads rl/models.py:456-506) as plain torch ops; keys as the reference's state_dict relative to one net.  Parity:
    pinned through tests/golden/variants_golden.npz (tests/test_oracle_model.py::test_mlp_forward_matches_reference)."""
    h = F.linear(torch.tanh(F.linear(x, sd["encoder.fc1.weight"], sd["encoder.fc1.bias"])),
                 sd["encoder.fc2.weight"], sd["encoder.fc2.bias"])
    f = torch.tanh(h) if activation == "tanh" else F.relu(h)
    out = {"raw_policy": F.linear(f, sd["policy_head.weight"], sd.get("policy_head.bias")),
           "value": F.linear(f, sd["value_head.weight"], sd.get("value_head.bias")),
           "advantage": F.linear(f, sd["advantage_head.weight"], sd.get("advantage_head.bias")), "h": h}
    if "tvf_head.weight" in sd:
        out["tvf_value"] = F.linear(f, sd["tvf_head.weight"], sd.get("tvf_head.bias"))
    return out


def ppo_loss(out, actions, old_log_pac, advantages, returns, eps=0.2, ent_coef=0.01, vf_coef=0.5, loss_scale=1.0):
    """mean((-gain) * loss_scale) exactly as rl/rollout.py:1640-1660,1682,1744-1753,1596-1608."""
    logps = out["log_policy"]
    B = logps.shape[0]
    logpac = logps[torch.arange(B), actions]
    ratio = torch.exp(logpac - old_log_pac)
    clipped = torch.clamp(ratio, 1 - eps, 1 + eps)
    loss_clip = torch.min(ratio * advantages, clipped * advantages)
    entropy = -(logps.exp() * logps).sum(-1)
    vloss = vf_coef * torch.square(out["value"][:, 0] - returns[:, 0])
    gain = loss_clip + ent_coef * entropy - vloss
    return ((-gain) * loss_scale).mean()


def forward_shared_kinks(sd, x, acts, n_stacks=3, n_block=2):
    """Same network, but every ReLU mask and max-pool selection is taken from `acts` (the HIP
    path's own saved pre-activations / argmax taps) instead of being re-decided.  In float64 this
    is the exact gradient of the function the HIP path evaluated: ReLU and max-pool are kinks
    where a 1e-7 forward difference flips a whole gradient entry, which says nothing about kernel
    arithmetic.  acts: dict with 'q{si}_{bi}_in', 'a{si}_{bi}', 'idx{si}', 'flat', 'h' tensors."""
    dt = x.dtype

    def relu_as(t, ref):
        return t * (ref > 0).to(dt)

    for si in range(n_stacks):
        p = f"encoder.stacks.{si}."
        c = F.conv2d(x, sd[p + "firstconv.weight"], sd[p + "firstconv.bias"], padding=1)
        idx = acts[f"idx{si}"].long()
        B, C, Ho, Wo = idx.shape
        H, W = c.shape[2], c.shape[3]
        oy = torch.arange(Ho, device=c.device).view(1, 1, Ho, 1)
        ox = torch.arange(Wo, device=c.device).view(1, 1, 1, Wo)
        pos = (2 * oy - 1 + idx // 3) * W + (2 * ox - 1 + idx % 3)
        x = c.flatten(2).gather(2, pos.flatten(2)).view(B, C, Ho, Wo)
        for bi in range(n_block):
            b = p + f"blocks.{bi}."
            r = F.conv2d(relu_as(x, acts[f"q{si}_{bi}_in"]), sd[b + "conv0.weight"], sd[b + "conv0.bias"], padding=1)
            r = F.conv2d(relu_as(r, acts[f"a{si}_{bi}"]), sd[b + "conv1.weight"], sd[b + "conv1.bias"], padding=1)
            x = x + r
    flat = x.reshape(x.shape[0], -1)
    h = F.linear(relu_as(flat, acts["flat"]), sd["encoder.dense.weight"], sd["encoder.dense.bias"])
    f = relu_as(h, acts["h"])
    raw = F.linear(f, sd["policy_head.weight"], sd.get("policy_head.bias"))
    return {"raw_policy": raw, "log_policy": F.log_softmax(raw, dim=1),
            "value": F.linear(f, sd["value_head.weight"], sd.get("value_head.bias")),
            "advantage": F.linear(f, sd["advantage_head.weight"], sd.get("advantage_head.bias")), "h": h}
