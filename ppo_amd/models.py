"""Policy/value network on hand-written HIP kernels — host mirror of the reference's
rl/models.py (TVFModel :511-856, DualHeadNet :304-508, ImpalaCNN :54-99) and rl/impala.py.

The network's arithmetic runs entirely in libppo_amd.so (f32-MFMA convolutions and GEMMs,
fused load transforms, fused PPO loss, fused Adam); this module owns the memory plan and the
call order.  Parameters live in ONE flat float32 device buffer (so the optimiser step and the
RCCL gradient all-reduce are single launches) and are exposed under the reference's
``state_dict`` names, so reference checkpoints load unchanged.

torch is used for device memory, views and (on the CPU, at construction only) the reference's
parameter initialisers; no torch operator is on the forward/backward path.
"""
import math
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib

IN_NONE, IN_RELU, IN_U8 = _lib.PPO_IN_NONE, _lib.PPO_IN_RELU, _lib.PPO_IN_U8
_ALIGN = 4  # floats: every parameter starts on a 16-byte boundary


def _p(t):
    return None if t is None else t.data_ptr()


# ----------------------------------------------------------------------------------------------
# parameter initialisation (host, CPU): the reference's initialisers in the reference's draw order
# ----------------------------------------------------------------------------------------------
def _normed_conv(cin, cout, scale=1.0):
    """rl/tensor_utilities.py:58-67 NormedConv2d: nn.Conv2d init, each filter L2-normalised * scale, zero bias."""
    m = torch.nn.Conv2d(cin, cout, 3, padding=1)
    with torch.no_grad():
        m.weight.data *= scale / m.weight.norm(dim=(1, 2, 3), p=2, keepdim=True)
        m.bias.data *= 0
    return m.weight.data, m.bias.data


def _normed_linear(fin, fout, scale=1.0):
    """rl/tensor_utilities.py:40-56 NormedLinear."""
    m = torch.nn.Linear(fin, fout)
    with torch.no_grad():
        m.weight.data *= scale / m.weight.norm(dim=1, p=2, keepdim=True)
        m.bias.data *= 0
    return m.weight.data, m.bias.data


def _custom_linear(fin, fout, scale=1.0, bias=True):
    """rl/tensor_utilities.py:69-86 CustomLinear(weight_init='orthogonal'): zero bias, orthogonal weight."""
    m = torch.nn.Linear(fin, fout, bias=bias)
    with torch.no_grad():
        if m.bias is not None:
            m.bias.data *= 0
        torch.nn.init.orthogonal_(m.weight.data, gain=scale)
    return m.weight.data, (m.bias.data if bias else None)


class ImpalaSpec:
    """Static geometry of the IMPALA encoder (rl/models.py:54-99, rl/impala.py:85-123)."""

    def __init__(self, input_dims, channels=(16, 32, 32), n_block=2, hidden_units=256):
        c, h, w = input_dims
        self.input_dims = tuple(input_dims)
        self.channels = tuple(channels)
        self.n_block = n_block
        self.hidden_units = hidden_units
        self.stacks = []  # (cin, cout, h_in, w_in, h_out, w_out)
        for cout in channels:
            ho, wo = (h + 1) // 2, (w + 1) // 2
            self.stacks.append((c, cout, h, w, ho, wo))
            c, h, w = cout, ho, wo
        self.out_shape = (c, h, w)
        self.flat = c * h * w


def init_impala_parameters(spec: ImpalaSpec, n_actions: int, vh: int, head_scale: float, head_bias: bool):
    """Initial parameters of DualHeadNet(encoder='impala') as CPU tensors, drawn from torch's global
    CPU generator in the reference's construction order (rl/models.py:348-368, :73-84; rl/impala.py:60-62,
    96-100), so that the same ``torch.manual_seed`` reproduces the reference's initial weights exactly.
    Keys are the reference's names relative to ``policy_net``."""
    init = OrderedDict()
    s_stack = 1 / math.sqrt(len(spec.channels))  # rl/models.py:75
    for si, (cin, cout, *_r) in enumerate(spec.stacks):
        w, b = _normed_conv(cin, cout)  # firstconv: scale 1 (rl/impala.py:96)
        init[f"encoder.stacks.{si}.firstconv.weight"], init[f"encoder.stacks.{si}.firstconv.bias"] = w, b
        s_block = math.sqrt(s_stack / math.sqrt(spec.n_block))  # rl/impala.py:60,97
        for bi in range(spec.n_block):
            for cname in ("conv0", "conv1"):
                w, b = _normed_conv(cout, cout, scale=s_block)
                init[f"encoder.stacks.{si}.blocks.{bi}.{cname}.weight"] = w
                init[f"encoder.stacks.{si}.blocks.{bi}.{cname}.bias"] = b
    w, b = _normed_linear(spec.flat, spec.hidden_units, scale=1.414)  # rl/models.py:84
    init["encoder.dense.weight"], init["encoder.dense.bias"] = w, b
    for name, rows in (("policy_head", n_actions), ("value_head", vh), ("advantage_head", n_actions)):
        w, b = _custom_linear(spec.hidden_units, rows, scale=head_scale, bias=head_bias)
        init[f"{name}.weight"] = w
        if head_bias:
            init[f"{name}.bias"] = b
    init["log_std"] = torch.zeros(n_actions)  # rl/models.py:368
    return init


class DualHeadNet:
    """One encoder + policy / value / advantage heads (reference: rl/models.py:304-508), HIP-backed.

    Only the IMPALA encoder is built on this path (``encoder='impala'``).
    """

    def __init__(self, encoder: str, input_dims, n_actions: int, hidden_units: int = 256,
                 activation_fn: str = "relu", head_scale: float = 1.0, value_head_names=("ext",),
                 head_bias: bool = False, device="cuda", **encoder_args):
        if encoder.lower() != "impala":
            raise NotImplementedError(f"encoder '{encoder}' has no HIP path yet (impala only)")
        if activation_fn != "relu":
            raise NotImplementedError("the impala path uses relu after the encoder (rl/train.py:67)")
        _lib.require_gpu()
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.PpoAmdError(f"device '{device}': the HIP path runs on the GPU only")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.spec = ImpalaSpec(input_dims, hidden_units=hidden_units, **encoder_args)
        self.n_actions = n_actions
        self.hidden_units = hidden_units
        self.value_head_names = list(value_head_names)
        self.vh = len(self.value_head_names)
        self.head_bias = head_bias
        self.nh = 2 * n_actions + self.vh  # fused head rows: policy | value | advantage
        self._build_parameters(head_scale)
        self._bufs: Dict[tuple, torch.Tensor] = {}
        self._adam_step = 0
        self.exp_avg = None
        self.exp_avg_sq = None

    # ------------------------------------------------------------------ parameters
    def _build_parameters(self, head_scale):
        sp = self.spec
        init = init_impala_parameters(sp, self.n_actions, self.vh, head_scale, self.head_bias)
        # physical order: encoder..., then the three head weights contiguous (one [nh, hidden] matrix),
        # then the three head biases contiguous, then log_std
        head_names = ("policy_head", "value_head", "advantage_head")
        order = [(n, t) for n, t in init.items() if n.startswith("encoder.")]
        order += [(f"{n}.weight", init[f"{n}.weight"]) for n in head_names]
        if self.head_bias:
            order += [(f"{n}.bias", init[f"{n}.bias"]) for n in head_names]
        order += [("log_std", init["log_std"])]

        offs, total = OrderedDict(), 0
        for name, t in order:
            offs[name] = (total, tuple(t.shape))
            n = t.numel()
            # head weights / biases must stay contiguous: no padding inside those groups
            contiguous_group = name.endswith("_head.weight") or name.endswith("_head.bias")
            total += n if contiguous_group else (n + _ALIGN - 1) // _ALIGN * _ALIGN
        total = (total + _ALIGN - 1) // _ALIGN * _ALIGN
        self.n_params_padded = total
        flat = torch.zeros(total, dtype=torch.float32)
        for name, t in order:
            o, shape = offs[name]
            flat[o:o + t.numel()] = t.reshape(-1)
        self.flat = flat.to(self.device)
        self.grad = torch.zeros_like(self.flat)
        self._offsets = offs
        self.params = OrderedDict((name, self.flat[o:o + int(np.prod(shape))].view(shape)) for name, (o, shape) in offs.items())
        self.grads = OrderedDict((name, self.grad[o:o + int(np.prod(shape))].view(shape)) for name, (o, shape) in offs.items())
        o = offs["policy_head.weight"][0]
        self.w_heads = self.flat[o:o + self.nh * sp.hidden_units].view(self.nh, sp.hidden_units)
        self.g_w_heads = self.grad[o:o + self.nh * sp.hidden_units].view(self.nh, sp.hidden_units)
        if self.head_bias:
            o = offs["policy_head.bias"][0]
            self.b_heads = self.flat[o:o + self.nh]
            self.g_b_heads = self.grad[o:o + self.nh]
        else:
            self.b_heads = self.g_b_heads = None

    def n_parameters(self) -> int:
        return sum(int(np.prod(s)) for _, s in self._offsets.values())

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        """Reference key names (policy_net.<these>), tensors are views into the flat buffer."""
        names = ["log_std"] + [n for n in self.params if n != "log_std"]
        return OrderedDict((n, self.params[n]) for n in names)

    def load_state_dict(self, sd, strict=True):
        missing = [n for n in self.params if n not in sd]
        unexpected = [n for n in sd if n not in self.params]
        if strict and (missing or unexpected):
            raise KeyError(f"state_dict mismatch: missing {missing}, unexpected {unexpected}")
        with torch.no_grad():
            for n, t in sd.items():
                if n in self.params:
                    self.params[n].copy_(torch.as_tensor(t).to(self.device, torch.float32).reshape(self.params[n].shape))

    # ------------------------------------------------------------------ scratch memory
    def _buf(self, name, shape, dtype=torch.float32):
        key = (name, tuple(shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t

    def _ws(self, name, nbytes):
        return self._buf(name, ((nbytes + 3) // 4,), torch.float32)

    def _call(self, fn_name, *args):
        rc = getattr(self.lib, fn_name)(*args, _lib.current_stream())
        if rc != 0:
            _lib.check(rc, fn_name)

    # ------------------------------------------------------------------ forward
    def _conv(self, x, mode, wname, out, residual, n, cin, cout, h, w):
        self._call("ppo_conv3x3_forward_f32", _p(x), mode, _p(self.params[wname + ".weight"]),
                   _p(self.params[wname + ".bias"]), _p(residual), _p(out), n, cin, cout, h, w)

    def encode(self, x: torch.Tensor, train: bool):
        """x: [B, C, H, W] uint8 (scaled /255 on load) or float32.  Returns the dict of saved
        pre-activation tensors; 'h' is the dense output before the encoder ReLU."""
        sp = self.spec
        B = x.shape[0]
        if tuple(x.shape[1:]) != sp.input_dims:
            raise ValueError(f"expected input [B, {sp.input_dims}], got {tuple(x.shape)}")
        if x.dtype not in (torch.uint8, torch.float32) or not x.is_contiguous() or x.device != self.device:
            raise ValueError("input must be a contiguous uint8/float32 tensor on the model's device")
        tag = "t" if train else "i"
        acts = {"x": x}
        cur, cur_mode = x, (IN_U8 if x.dtype == torch.uint8 else IN_NONE)
        for si, (cin, cout, h, w, ho, wo) in enumerate(sp.stacks):
            c = self._buf(f"{tag}c{si}", (B, cout, h, w))
            self._conv(cur, cur_mode, f"encoder.stacks.{si}.firstconv", c, None, B, cin, cout, h, w)
            p = self._buf(f"{tag}p{si}", (B, cout, ho, wo))
            idx = self._buf(f"{tag}idx{si}", (B, cout, ho, wo), torch.uint8) if train else None
            self._call("ppo_maxpool3x3s2_forward_f32", _p(c), _p(p), _p(idx), B, cout, h, w)
            acts[f"in{si}"], acts[f"idx{si}"] = cur, idx
            q = p
            for bi in range(sp.n_block):
                a = self._buf(f"{tag}a{si}_{bi}", (B, cout, ho, wo))
                qn = self._buf(f"{tag}q{si}_{bi}", (B, cout, ho, wo))
                base = f"encoder.stacks.{si}.blocks.{bi}"
                self._conv(q, IN_RELU, base + ".conv0", a, None, B, cout, cout, ho, wo)
                self._conv(a, IN_RELU, base + ".conv1", qn, q, B, cout, cout, ho, wo)
                acts[f"q{si}_{bi}_in"], acts[f"a{si}_{bi}"] = q, a
                q = qn
            cur, cur_mode = q, IN_NONE
        flat = cur.view(B, sp.flat)
        h = self._buf(f"{tag}h", (B, sp.hidden_units))
        wd = self.params["encoder.dense.weight"]
        ws_bytes = self.lib.ppo_gemm_workspace_bytes(B, sp.hidden_units, sp.flat)
        ws = self._ws("gemm_ws", ws_bytes)
        self._call("ppo_gemm_f32", _p(flat), sp.flat, 1, 1, _p(wd), 1, sp.flat, 0,
                   _p(self.params["encoder.dense.bias"]), None, _p(h), sp.hidden_units, B, sp.hidden_units, sp.flat,
                   _p(ws), ws_bytes)
        acts["flat"], acts["h"] = flat, h
        return acts

    def heads(self, h: torch.Tensor, tag="i"):
        """[B, nh] = relu(h) @ [policy | value | advantage]^T (+ bias)  (rl/models.py:467-506)."""
        B = h.shape[0]
        o = self._buf(f"{tag}heads", (B, self.nh))
        self._call("ppo_gemm_f32", _p(h), self.hidden_units, 1, 1, _p(self.w_heads), 1, self.hidden_units, 0,
                   _p(self.b_heads), None, _p(o), self.nh, B, self.nh, self.hidden_units, None, 0)
        return o

    def forward(self, x, policy_temperature: float = 1.0, train: bool = False) -> Dict[str, torch.Tensor]:
        """Same result keys as the reference's DualHeadNet.forward (rl/models.py:433-508)."""
        acts = self.encode(x, train)
        o = self.heads(acts["h"], "t" if train else "i")
        B, nA = o.shape[0], self.n_actions
        logp = self._buf("log_policy", (B, nA))
        result = {"raw_policy": o[:, :nA], "value": o[:, nA:nA + self.vh], "advantage": o[:, nA + self.vh:]}
        if policy_temperature > 0:
            self._call("ppo_policy_act_f32", _p(o), B, self.nh, nA, float(policy_temperature), None, 0, 0, 1, _p(logp),
                       None, None, None, None, self.vh)
            result["log_policy"] = logp
        else:
            # greedy / blended policy (rl/models.py:475-485): tiny [B, nA] tensors, composed from the
            # HIP log-softmax and argmax
            act = self._buf("greedy_actions", (B,), torch.int32)
            self._call("ppo_policy_act_f32", _p(o), B, self.nh, nA, 1.0, None, 0, 0, 1, _p(logp), _p(act), None,
                       None, None, self.vh)
            argmax_policy = torch.zeros_like(logp)
            argmax_policy[torch.arange(B, device=self.device), act.long()] = 1.0
            eps = 1 + policy_temperature
            result["log_policy"] = torch.log(eps * argmax_policy + (1 - eps) * torch.exp(logp) + 1e-8)
            result["argmax_policy"] = argmax_policy
        result["_heads"], result["_acts"] = o, acts
        return result

    # ------------------------------------------------------------------ backward
    def backward(self, acts, dheads: torch.Tensor, accumulate: bool = False):
        """Back-propagate d loss / d heads through heads, dense layer and encoder into self.grad."""
        sp, lib = self.spec, self.lib
        B = dheads.shape[0]
        H = sp.hidden_units
        acc = 1 if accumulate else 0
        if accumulate:
            raise NotImplementedError("gradient accumulation over micro-batches is not wired for the GEMM layers yet")
        h, flat = acts["h"], acts["flat"]
        # heads: dW = dheads^T @ relu(h); db = colsum(dheads); dh = (dheads @ W) * (h > 0)
        self._call("ppo_gemm_f32", _p(dheads), 1, self.nh, 0, _p(h), H, 1, 1, None, None, _p(self.g_w_heads), H,
                   self.nh, H, B, None, 0)
        if self.head_bias:
            self._call("ppo_colsum_f32", _p(dheads), B, self.nh, self.nh, _p(self.g_b_heads), acc)
        dh = self._buf("dh", (B, H))
        self._call("ppo_gemm_f32", _p(dheads), self.nh, 1, 0, _p(self.w_heads), H, 1, 0, None, _p(h), _p(dh), H, B, H,
                   self.nh, None, 0)
        # dense: dW = dh^T @ relu(flat); db = colsum(dh); dflat = (dh @ W) * (flat > 0)
        wd = self.params["encoder.dense.weight"]
        self._call("ppo_gemm_f32", _p(dh), 1, H, 0, _p(flat), sp.flat, 1, 1, None, None,
                   _p(self.grads["encoder.dense.weight"]), sp.flat, H, sp.flat, B, None, 0)
        self._call("ppo_colsum_f32", _p(dh), B, H, H, _p(self.grads["encoder.dense.bias"]), acc)
        c_last, h_last, w_last = sp.out_shape
        g = self._buf(f"g{len(sp.stacks) - 1}_a", (B, c_last, h_last, w_last))
        self._call("ppo_gemm_f32", _p(dh), H, 1, 0, _p(wd), sp.flat, 1, 0, None, _p(flat), _p(g), sp.flat, B, sp.flat,
                   H, None, 0)

        def wgrad(x, mode, dy, wname, n, cin, cout, hh, ww):
            nbytes = lib.ppo_conv3x3_wgrad_workspace_bytes(cin, cout)
            ws = self._ws("wgrad_ws", nbytes)
            self._call("ppo_conv3x3_backward_weight_f32", _p(x), mode, _p(dy), _p(self.grads[wname + ".weight"]),
                       _p(self.grads[wname + ".bias"]), _p(ws), nbytes, n, cin, cout, hh, ww, acc)

        for si in reversed(range(len(sp.stacks))):
            cin, cout, hh, ww, ho, wo = sp.stacks[si]
            slot = 0  # g currently lives in buffer g{si}_a
            for bi in reversed(range(sp.n_block)):
                base = f"encoder.stacks.{si}.blocks.{bi}"
                q_in, a = acts[f"q{si}_{bi}_in"], acts[f"a{si}_{bi}"]
                # g = d loss / d (block output);  block: out = q_in + conv1(relu(conv0(relu(q_in))))
                wgrad(a, IN_RELU, g, base + ".conv1", B, cout, cout, ho, wo)
                da = self._buf(f"g{si}_da", (B, cout, ho, wo))
                self._call("ppo_conv3x3_backward_data_f32", _p(g), _p(self.params[base + ".conv1.weight"]), _p(a), None,
                           _p(da), B, cout, cout, ho, wo)
                wgrad(q_in, IN_RELU, da, base + ".conv0", B, cout, cout, ho, wo)
                slot ^= 1  # ping-pong between the two gradient buffers of this resolution
                gn = self._buf(f"g{si}_{'ab'[slot]}", (B, cout, ho, wo))
                self._call("ppo_conv3x3_backward_data_f32", _p(da), _p(self.params[base + ".conv0.weight"]), _p(q_in),
                           _p(g), _p(gn), B, cout, cout, ho, wo)
                g = gn
            # max-pool backward, then the stack's first convolution
            dc = self._buf(f"g{si}_dc", (B, cout, hh, ww))
            self._call("ppo_maxpool3x3s2_backward_f32", _p(g), _p(acts[f"idx{si}"]), _p(dc), B, cout, hh, ww)
            x_in = acts[f"in{si}"]
            mode = IN_NONE if si > 0 else (IN_U8 if x_in.dtype == torch.uint8 else IN_NONE)
            wgrad(x_in, mode, dc, f"encoder.stacks.{si}.firstconv", B, cin, cout, hh, ww)
            if si > 0:
                g = self._buf(f"g{si - 1}_a", (B, cin, hh, ww))
                self._call("ppo_conv3x3_backward_data_f32", _p(dc), _p(self.params[f"encoder.stacks.{si}.firstconv.weight"]),
                           None, None, _p(g), B, cin, cout, hh, ww)

    # ------------------------------------------------------------------ PPO minibatch + optimiser
    def ppo_minibatch(self, prev_state, actions, old_log_pac, old_log_policy, advantages, returns,
                      eps_clip=0.2, ent_coef=0.01, vf_coef=0.5, loss_scale=1.0, index=None):
        """Forward, fused PPO loss, backward: gradients of mean(-gain)*loss_scale land in self.grad
        (Runner.train_policy_minibatch, rl/rollout.py:1610-1771, single architecture).
        prev_state is the (already gathered) minibatch of observations; the per-sample arrays are
        either minibatch-sized or, with ``index`` ([B] int32), whole-batch arrays read at index[b].
        Returns the per-sample statistics tensor [B, 8] (device)."""
        acts = self.encode(prev_state, train=True)
        o = self.heads(acts["h"], "t")
        B = o.shape[0]
        dheads = self._buf("dheads", (B, self.nh))
        stats = self._buf("loss_stats", (B, 8))
        self._call("ppo_ppo_loss_f32", _p(o), B, self.nh, self.n_actions, self.vh, _p(actions), _p(old_log_pac),
                   _p(old_log_policy), _p(advantages), _p(returns), float(eps_clip), float(ent_coef), float(vf_coef),
                   float(loss_scale) / B, _p(dheads), _p(stats), _p(index))
        self.backward(acts, dheads)
        return stats

    def adam_step(self, lr=2.5e-4, beta1=0.9, beta2=0.999, eps=1e-5, max_grad_norm=20.0, grad_div=1.0,
                  grad_norm_out: Optional[torch.Tensor] = None):
        """clip_grad_norm_ + torch.optim.Adam.step over the flat buffer (rl/rollout.py:1287-1321)."""
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.flat)
            self.exp_avg_sq = torch.zeros_like(self.flat)
        self._adam_step += 1
        ws = self._ws("adam_ws", self.lib.ppo_adam_workspace_bytes())
        self._call("ppo_adam_step_f32", _p(self.flat), _p(self.grad), _p(self.exp_avg), _p(self.exp_avg_sq),
                   self.flat.numel(), self._adam_step, float(lr), float(beta1), float(beta2), float(eps),
                   float(max_grad_norm), float(grad_div), _p(ws), _p(grad_norm_out))

    # ------------------------------------------------------------------ optimiser state (checkpoints)
    def optimizer_state_dict(self):
        """Adam state in torch.optim.Adam's layout idea (step + per-parameter exp_avg / exp_avg_sq),
        keyed by parameter name (rl/rollout.py:394-453 stores optimizer.state_dict())."""
        if self.exp_avg is None:
            return {"step": 0, "state": {}}
        state = {}
        for name, (o, shape) in self._offsets.items():
            n = int(np.prod(shape))
            state[name] = {"exp_avg": self.exp_avg[o:o + n].view(shape).clone(),
                           "exp_avg_sq": self.exp_avg_sq[o:o + n].view(shape).clone()}
        return {"step": self._adam_step, "state": state}

    def load_optimizer_state_dict(self, sd):
        self._adam_step = int(sd.get("step", 0))
        if not sd.get("state"):
            self.exp_avg = self.exp_avg_sq = None
            return
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        for name, st in sd["state"].items():
            o, shape = self._offsets[name]
            n = int(np.prod(shape))
            self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))


class TVFModel:
    """Host mirror of the reference's TVFModel (rl/models.py:511-856) for the PPO path: owns
    `policy_net` (and, for `architecture='single'`, `value_net is policy_net`), exposes
    `forward(x, output=..., policy_temperature=...) -> dict` with the reference's key aliasing
    (:790-796) and a `state_dict` with the reference's `policy_net.` / `value_net.` prefixes."""

    def __init__(self, encoder: str, encoder_args=None, input_dims=(4, 84, 84), actions: int = 6, device="cuda",
                 architecture: str = "dual", dtype=torch.float32, use_rnd: bool = False, hidden_units: int = 512,
                 encoder_activation_fn: str = "relu", observation_normalization=False,
                 freeze_observation_normalization=False, tvf_fixed_head_horizons=None, tvf_fixed_head_weights=None,
                 tvf_feature_sparsity: float = 0.0, tvf_feature_window: int = -1, head_scale: float = 1.0,
                 value_head_names=("ext",), norm_eps: float = 1e-5, head_bias: bool = False,
                 observation_scaling: str = "scaled"):
        if architecture != "single":
            raise NotImplementedError("HIP path: architecture='single' (PPO). 'dual' (DNA) is listed as next in DESIGN.md")
        if use_rnd or observation_normalization or tvf_fixed_head_horizons is not None:
            raise NotImplementedError("RND / observation normalisation / TVF heads are not on the HIP path yet")
        if dtype != torch.float32:
            raise ValueError("the reference path is float32 (rl/models.py:31-32)")
        if observation_scaling != "scaled":
            raise NotImplementedError("observation_scaling='scaled' only (x/255 fused into the first conv)")
        if isinstance(encoder_args, str):
            import ast
            encoder_args = ast.literal_eval(encoder_args)
        self.input_dims = tuple(input_dims)
        self.actions = actions
        self.device = device
        self.dtype = dtype
        self.architecture = architecture
        self.encoder_name = encoder
        self.name = "PPO-" + encoder
        self.policy_net = DualHeadNet(encoder, input_dims, actions, hidden_units=hidden_units,
                                      activation_fn=encoder_activation_fn, head_scale=head_scale,
                                      value_head_names=value_head_names, head_bias=head_bias, device=device,
                                      **(encoder_args or {}))
        self.value_net = self.policy_net
        self.device = self.policy_net.device

    def model_size(self, trainable_only: bool = True):
        return self.policy_net.n_parameters()

    def prep_for_model(self, x):
        """rl/models.py:824-856: accept ndarray or tensor, uint8 or float; the /255 scaling itself is fused
        into the first convolution's load."""
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        if x.dtype not in (torch.uint8, torch.float32):
            raise AssertionError("Invalid dtype {}".format(x.dtype))
        if tuple(x.shape[1:]) != self.input_dims:
            raise AssertionError("Invalid dims, expected {} but found {}".format((None, *self.input_dims), tuple(x.shape)))
        return x.to(self.device, non_blocking=True).contiguous()

    def forward(self, x, output: str = "default", policy_temperature: float = 1.0, include_rnd=False,
                include_features=False, update_normalization=False, **kwargs):
        assert output in ["default", "full", "policy", "value"]
        out = self.policy_net.forward(self.prep_for_model(x), policy_temperature=policy_temperature)
        result = {}
        for k, v in out.items():
            if k.startswith("_"):
                continue
            result["policy_" + k] = v
            result["value_" + k] = v
            result[k] = v
        return result

    __call__ = forward

    def log_policy(self, x):
        return self.forward(x, output="policy")["log_policy"].detach().cpu().numpy()

    def state_dict(self):
        sd = OrderedDict()
        for prefix in ("policy_net.", "value_net."):
            for k, v in self.policy_net.state_dict().items():
                sd[prefix + k] = v
        return sd

    def load_state_dict(self, sd, strict=True):
        pol = {k[len("policy_net."):]: v for k, v in sd.items() if k.startswith("policy_net.")}
        self.policy_net.load_state_dict(pol, strict=strict)
