"""Policy/value networks on hand-written HIP kernels — host mirror of the reference's
rl/models.py (TVFModel :511-856, DualHeadNet :304-508, ImpalaCNN :54-99, StandardMLP :148-169) and
rl/impala.py.

The network's arithmetic runs entirely in libppo_amd.so (f32-MFMA convolutions and GEMMs,
fused load transforms, fused losses, fused Adam); this module owns the memory plan and the
call order.  Parameters of one net live in ONE flat float32 device buffer (so the optimiser step
and the RCCL gradient all-reduce are single launches) and are exposed under the reference's
``state_dict`` names, so reference checkpoints load unchanged.

torch is used for device memory, views and (on the CPU, at construction only) the reference's
parameter initialisers; no torch operator is on the forward/backward path.
"""
import ctypes
import math
import os
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib

IN_NONE, IN_RELU, IN_U8 = _lib.PPO_IN_NONE, _lib.PPO_IN_RELU, _lib.PPO_IN_U8
_ALIGN = 4  # floats: every parameter starts on a 16-byte boundary
# bit i set: stack i runs its first convolution fused with the max-pool (bit-identical either way; the choice is
# a measured one)
FUSE_POOL_STACKS = int(os.environ.get("PPO_AMD_FUSE_POOL", "7"))
# weight-gradient kernels on a second stream, overlapping the backward-data chain (0 = one stream).  It paid while single
# weight-gradient launches left the chip half empty (round 1: -4 %); with the batched, one-wave launches every kernel
# fills the chip on its own and the second stream costs ~1 % (83.1 k vs 82.3 k env-steps/s), so the default is one stream.
WGRAD_SIDE_STREAM = int(os.environ.get("PPO_AMD_WGRAD_STREAM", "0"))
# the slab reductions of all convolution layers in one launch at the end of the backward pass (0 = one per layer)
WGRAD_BATCH_REDUCE = int(os.environ.get("PPO_AMD_WGRAD_BATCH_REDUCE", "1"))
# The four block convolutions of a stack share a geometry: their weight gradients go out as ONE launch (4 x the
# workgroups, one ramp and one tail) once the stack's backward-data pass is through.  Needs WGRAD_BATCH_REDUCE.
WGRAD_BATCH_LAUNCH = int(os.environ.get("PPO_AMD_WGRAD_BATCH_LAUNCH", "1"))
# The first stack's max-pool backward is folded into its first convolution's weight-gradient kernel (the only reader of
# that 84x84 gradient map: nothing back-propagates into the observations).  Needs WGRAD_BATCH_REDUCE.  Measured: the
# max-pool backward launch (37 us) goes, the weight-gradient kernel grows by ~20 us (it gathers (argmax, g) pairs
# instead of streaming the map in by LDS-DMA): -15 us per step net and a 115 MB tensor less.
WGRAD_POOLED_DY = int(os.environ.get("PPO_AMD_WGRAD_POOLED_DY", "1"))
# a stack's first convolution whose geometry equals the PREVIOUS stack's block convolutions (32 -> 32 at 21x21 for the 84x84
# net) has its weight gradient formed by that stack's batched launch, as a fifth problem read without the ReLU
# (ppo_conv3x3_backward_weight_slabs_batch_mixed_f32): one launch less per backward pass (11 - 14 us of fixed cost)
WGRAD_RIDE = int(os.environ.get("PPO_AMD_WGRAD_RIDE", "1"))
# --precision=medium|low only: weight gradients of the 16- and 32-channel float layers as split-bf16 products too
# (csrc/wgrad_bf16x3.hip; 0 keeps them on the exact float32 kernel while the residual blocks stay split).
SPLIT_WGRAD = int(os.environ.get("PPO_AMD_SPLIT_WGRAD", "1"))
# likewise the stack-first convolutions (forward: split convolution + the max-pool launch instead of the fused float32
# conv + pool kernel; backward-data): csrc/conv_bf16x3.hip
SPLIT_CONV = int(os.environ.get("PPO_AMD_SPLIT_CONV", "1"))
SPLIT_CONV_POOL = int(os.environ.get("PPO_AMD_SPLIT_CONV_POOL", "1"))  # ... with the max-pool inside the launch
SPLIT_SIGNS = int(os.environ.get("PPO_AMD_SPLIT_SIGNS", "1"))  # the split backward chains read 1-bit gates written by the forward
# convolutions read their MFMA A operand from a pre-packed copy of the weights, refreshed by one launch after every
# optimiser step (0 = every kernel stages the raw tensor through LDS itself); bit-identical either way
PACKED_WEIGHTS = int(os.environ.get("PPO_AMD_PACKED_WEIGHTS", "1"))
# the optimiser step writes the updated convolution weights into their packed layouts too (no re-pack launch per step).
# Measured and NOT kept (0): 1.118 ms per training step against 1.094 with the separate 5 us pack launch - 2 x 98 k
# scattered 4-byte writes from the first hundred workgroups of the Adam launch cost more than the coalesced re-pack
ADAM_SCATTER = int(os.environ.get("PPO_AMD_ADAM_SCATTER", "0"))
# The two residual blocks of a stack as one launch with the image resident in LDS (csrc/stack_fused.hip), where the
# geometry has a kernel (32 channels at 11x11) and the packed weights exist; 0 = four convolution launches.
FUSE_STACK_TAIL = int(os.environ.get("PPO_AMD_FUSE_STACK_TAIL", "1"))
# A whole stack (first convolution + max-pool + both blocks) in one launch where its input map and pre-pool map fit LDS
# together (32 channels at 21x21 -> 11x11, i.e. the last stack of the 84x84 net); 0 = conv+pool launch, then the tail.
FUSE_STACK_FULL = int(os.environ.get("PPO_AMD_FUSE_STACK_FULL", "1"))
# ... with the previous stack's residual blocks chained in front of it (that stack's output stays in LDS).
FUSE_STACK_CHAIN = int(os.environ.get("PPO_AMD_FUSE_STACK_CHAIN", "1"))
# inference batches of at most CHAIN_SPLIT_MAX_BATCH images (a rollout group) run the chained launch with every image
# on TWO workgroups (output channels split, halves exchanged per layer): same bits, the whole chip instead of half of it.
# Alone, a 128-image forward takes 0.258 ms against 0.282 (the launch 104 us against 128).  In the two-group rollout an
# env step is one group's forward latency plus its host leg, so the shorter launch pays even though it spends 1.6 x
# the CU time: variants alternated rollout by rollout in one process (tools/rollout_ab.py; separate processes differ by
# more than the effect) 0.4846 -> 0.4649 ms per env step, 0.4584 with the action step inside the heads launch as well.
# A partner that never arrives sets an error word which the Runner checks after every rollout (chain_split_error) and
# answers by dropping to the one-workgroup launch and redoing the rollout (Runner.generate_rollout).  Only a forward made
# under `net.allow_chain_split` (the Runner's pipelined rollout sets it, and checks) takes the split launch: evaluation,
# greedy and generic-rollout forwards, whose callers never look at the error word, keep the one-workgroup launch.
CHAIN_SPLIT = int(os.environ.get("PPO_AMD_CHAIN_SPLIT", "1"))
CHAIN_SPLIT_MAX_BATCH = int(os.environ.get("PPO_AMD_CHAIN_SPLIT_MAX_BATCH", "128"))
# ... and its backward-data pass (blocks + max-pool backward + transposed first convolution) likewise.  Off by default:
# bit-identical, but 1.445 ms per 256-sample step against 1.395 without it — it holds a whole CU's LDS, so the
# weight-gradient kernels on the side stream get nothing to overlap with for its duration, and releases the five
# gradients they wait for only at its end.
FUSE_STACK_FULL_BWD = int(os.environ.get("PPO_AMD_FUSE_STACK_FULL_BWD", "0"))
# ... and the same for their backward-data chain, as a bit mask over the stacks (per 256-sample step, same box:
# 1.476 ms with mask 0, 1.450 with 4 (11x11), 1.404 with 2 (21x21), 1.408 with 6).
FUSE_STACK_TAIL_BWD = int(os.environ.get("PPO_AMD_FUSE_STACK_TAIL_BWD", "7"))
# The 16-channel stack's blocks (42x42 / 32x32: the map fills most of a CU's LDS) as the in-place, row-shifted form of the
# same kernel family (csrc/stack_fused.hip stack_shift_kernel); 0 = four convolution launches.  One 16-wave workgroup per
# image and per CU, two wave groups half a band apart (one in its K loop while the other runs its epilogue): measured at
# batch 256 forward 0.398 -> 0.393 ms, training step 1.219 -> 1.186 ms with the backward-data form too (bit 0 of the mask
# above).  At 128 images (a rollout group) it leaves half the chip idle (0.296 -> 0.325 ms), so batches below
# FUSE_STACK16_MIN_BATCH keep the four launches.  Same bits either way.
FUSE_STACK16 = int(os.environ.get("PPO_AMD_FUSE_STACK16", "1"))
FUSE_STACK16_MIN_BATCH = int(os.environ.get("PPO_AMD_FUSE_STACK16_MIN_BATCH", "192"))
# Below that batch an inference forward runs each 16-channel residual block as one launch (csrc/conv3x3_block.hip: band by
# band, the intermediate map in LDS) instead of two convolution launches: a 128-image group is launch-cost-bound there.
FUSE_BLOCK = int(os.environ.get("PPO_AMD_FUSE_BLOCK", "1"))
# MLP nets (encoder "mlp": the continuous-control and classic configs) run as fused launches (csrc/mlp_fused.hip): one
# per inference forward, three per training minibatch (rows kernel: forward + loss + backward-data; weight gradients +
# statistics; Adam) instead of ~14 launches of 8 - 20 us each.  0 = the op-by-op path (same arithmetic per element up to
# float32 summation order; tests/test_variants_gpu.py runs both against the reference's fixtures).
FUSE_MLP = int(os.environ.get("PPO_AMD_FUSE_MLP", "1"))
# the discrete PPO loss runs on the finished head row inside the training forward's dense + heads launch
# (ppo_dense_heads_loss_forward_f32): bit-identical, one latency-bound launch less per minibatch
FUSE_LOSS = int(os.environ.get("PPO_AMD_FUSE_LOSS", "1"))
# uint8 image minibatches are read out of the whole rollout batch through the permutation by the first convolution (and by
# its weight gradient) instead of being gathered into a second buffer by a launch of its own (0 = gather first)
GATHER_IN_CONV = int(os.environ.get("PPO_AMD_GATHER_IN_CONV", "1"))
HEAD_NAMES = ("policy_head", "value_head", "advantage_head", "tvf_head")


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
    kind = "impala"

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


class MLPSpec:
    """StandardMLP (rl/models.py:148-169): fc1 -> tanh -> fc2 on a flat float observation."""
    kind = "mlp"

    def __init__(self, input_dims, hidden_units=64):
        if len(input_dims) != 1:
            raise ValueError(f"the mlp encoder takes flat observations, got input_dims={input_dims}")
        self.input_dims = tuple(input_dims)
        self.in_features = int(input_dims[0])
        self.hidden_units = hidden_units


def init_encoder_parameters(spec):
    """Encoder parameters as CPU tensors, drawn from torch's global CPU generator in the reference's
    construction order (impala: rl/models.py:73-84, rl/impala.py:60-62, 96-100; mlp: rl/models.py:157-161)."""
    init = OrderedDict()
    if spec.kind == "mlp":
        w, b = _custom_linear(spec.in_features, spec.hidden_units, scale=torch.nn.init.calculate_gain("tanh"))
        init["encoder.fc1.weight"], init["encoder.fc1.bias"] = w, b
        w, b = _custom_linear(spec.hidden_units, spec.hidden_units, scale=1.414)
        init["encoder.fc2.weight"], init["encoder.fc2.bias"] = w, b
        return init
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
    return init


def init_parameters(spec, n_actions: int, vh: int, head_scale: float, head_bias: bool, n_tvf: int = 0):
    """Initial parameters of DualHeadNet as CPU tensors in the reference's construction order
    (rl/models.py:348-384: encoder, policy / value / advantage heads, log_std, then the TVF head), so
    that the same ``torch.manual_seed`` reproduces the reference's initial weights exactly.
    Keys are the reference's names relative to ``policy_net``."""
    init = init_encoder_parameters(spec)
    heads = [("policy_head", n_actions), ("value_head", vh), ("advantage_head", n_actions)]
    if n_tvf:
        heads.append(("tvf_head", n_tvf * vh))
    for name, rows in heads:
        w, b = _custom_linear(spec.hidden_units, rows, scale=head_scale, bias=head_bias)
        init[f"{name}.weight"] = w
        if head_bias:
            init[f"{name}.bias"] = b
    init["log_std"] = torch.zeros(n_actions)  # rl/models.py:368 (no random draw, so its position is free)
    return init


def init_impala_parameters(spec: ImpalaSpec, n_actions: int, vh: int, head_scale: float, head_bias: bool):
    return init_parameters(spec, n_actions, vh, head_scale, head_bias)


def tvf_feature_mask(K: int, H: int, sparsity: float, window: int) -> torch.Tensor:
    """The scaled mask [K, H] of rl/models.py:386-421 (CPU tensor): `sparsity` keeps each (head, feature) with probability
    1 - sparsity and scales the kept ones by sqrt(1 / keep) - drawn from a CPU generator seeded 99, i.e. what the reference
    draws on --device=cpu (torch's CPU generators give the same stream only on hosts that take the same vector code path);
    `window` gives head k the features [left_k, right_k), the window sliding from the first to the last feature with k,
    scaled by sqrt(H / window)."""
    mask = torch.ones([K, H], dtype=torch.float32)
    if sparsity > 0:
        keep_prob = 1 - sparsity
        g = torch.Generator(device="cpu")
        g.manual_seed(99)
        scaled = torch.bernoulli(mask * keep_prob, generator=g) * math.sqrt(1 / keep_prob)
    if window > 0:
        assert sparsity <= 0, "sparsity and feature window not supported together"
        first_right, last_left = window, H - window
        for head in range(K):
            factor = head / (K - 1)
            left = int(0 * (1 - factor) + last_left * factor)
            right = int(first_right * (1 - factor) + H * factor)
            mask[head, :left] = 0
            mask[head, right:] = 0
        scaled = mask * ((1 / math.sqrt(window)) / (1 / math.sqrt(H)))
    return scaled


def plan_input_keys(kind):
    """Entries of encode()'s result that alias the input tensor."""
    return ("x", "in0") if kind == "impala" else ("x",)


class AdamState:
    """Adam moments + step count over a net's flat parameter buffer."""

    def __init__(self):
        self.exp_avg = self.exp_avg_sq = None
        self.step = 0

    def ensure(self, flat):
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(flat)
            self.exp_avg_sq = torch.zeros_like(flat)


class ObsNormalizer:
    """Device-resident form of the reference's observation normaliser (rl/models.py:615-617, 661-694):
    `obs_rms` (utils.RunningMeanStd: float64 mean / var per feature, scalar count) plus the float32 constants
    `_mu` / `_std`.  One instance is shared by the policy and value nets of a TVFModel; each net applies it in
    `encode`.  Data-parallel ranks all-reduce the batch moments, so every rank holds the same statistics."""

    def __init__(self, input_dims, device, norm_eps: float = 1e-5, frozen: bool = False, epsilon: float = 1e-4):
        self.input_dims = tuple(int(d) for d in input_dims)
        self.F = int(np.prod(self.input_dims))
        self.device = torch.device(device)
        self.norm_eps, self.frozen = float(norm_eps), bool(frozen)
        self.mean = torch.zeros(self.F, dtype=torch.float64, device=self.device)
        self.var = torch.ones(self.F, dtype=torch.float64, device=self.device)
        self.count = float(epsilon)
        self.mu = torch.zeros(self.F, dtype=torch.float32, device=self.device)
        self.std = torch.ones(self.F, dtype=torch.float32, device=self.device)
        self._moments = torch.empty(2 * self.F, dtype=torch.float64, device=self.device)
        self.device = self.mean.device  # 'cuda' -> 'cuda:0', comparable with tensors' devices
        self.lib = _lib.load()

    def _call(self, fn_name, *args):
        rc = getattr(self.lib, fn_name)(*args, _lib.current_stream())
        if rc != 0:
            _lib.check(rc, fn_name)

    def _check(self, x):
        if tuple(x.shape[1:]) != self.input_dims or x.dtype not in (torch.uint8, torch.float32) \
                or not x.is_contiguous() or x.device != self.device:
            raise ValueError(f"expected a contiguous uint8/float32 [B, {self.input_dims}] tensor on {self.device}")

    def update(self, x: torch.Tensor):
        """perform_normalization(..., update_normalization=True)'s statistics update (rl/models.py:681-687)."""
        if self.frozen or x.shape[0] == 0:
            return
        self._check(x)
        from . import parallel
        self._call("ppo_obs_moments_f64", _p(x), 1 if x.dtype == torch.uint8 else 0, x.shape[0], self.F,
                   _p(self._moments))
        n = x.shape[0]
        if parallel.world_size() > 1:
            parallel.allreduce_sum_(self._moments)
            n *= parallel.world_size()
        self._call("ppo_obs_rms_update_f64", _p(self._moments), float(n), self.count, _p(self.mean), _p(self.var),
                   _p(self.mu), _p(self.std), self.F)
        self.count += n

    def apply(self, x: torch.Tensor, out: torch.Tensor):
        self._check(x)
        self._call("ppo_obs_normalize_f32", _p(x), 1 if x.dtype == torch.uint8 else 0, _p(self.mu), _p(self.std),
                   self.norm_eps, _p(out), x.shape[0], self.F)
        return out

    def state_dict(self):
        return {"mean": self.mean.cpu(), "var": self.var.cpu(), "count": float(self.count)}

    def load_state_dict(self, sd):
        self.mean.copy_(torch.as_tensor(sd["mean"], dtype=torch.float64).reshape(self.F))
        self.var.copy_(torch.as_tensor(sd["var"], dtype=torch.float64).reshape(self.F))
        self.count = float(sd["count"])
        self.mu.copy_(self.mean.float())                # refresh_normalization_constants, rl/models.py:661-663
        self.std.copy_(self.var.float().sqrt())


class DualHeadNet:
    """One encoder + policy / value / advantage (/ TVF) heads (reference: rl/models.py:304-508), HIP-backed.

    Encoders: ``impala`` (3x3 conv stacks, uint8 or float images) and ``mlp`` (flat float observations);
    encoder activation ``relu`` (fused into the head GEMM's operand load) or ``tanh``.  All heads are ONE
    [nh, hidden] matrix so a forward is one GEMM: columns [policy nA | value VH | advantage nA | tvf K*VH].
    """

    def __init__(self, encoder: str, input_dims, n_actions: int, hidden_units: int = 256,
                 activation_fn: str = "relu", tvf_fixed_head_horizons=None, tvf_feature_sparsity: float = 0.0,
                 tvf_feature_window: int = -1, head_scale: float = 1.0, value_head_names=("ext",),
                 head_bias: bool = False, device="cuda", precision: str = "high", **encoder_args):
        encoder = encoder.lower()
        if precision not in ("low", "medium", "high"):
            raise ValueError(f"Invalid precision mode {precision}")
        if encoder not in ("impala", "mlp"):
            raise NotImplementedError(f"encoder '{encoder}' has no HIP path (impala | mlp)")
        if activation_fn not in ("relu", "tanh"):
            raise ValueError(f"Invalid activation function {activation_fn}")
        _lib.require_gpu()
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.PpoAmdError(f"device '{device}': the HIP path runs on the GPU only")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if encoder == "impala":
            self.spec = ImpalaSpec(input_dims, hidden_units=hidden_units, **encoder_args)
        else:
            self.spec = MLPSpec(input_dims, hidden_units=hidden_units, **encoder_args)
        self.encoder_kind = encoder
        self.encoder_activation_fn = activation_fn
        self.n_actions = n_actions
        self.hidden_units = hidden_units
        self.value_head_names = list(value_head_names)
        self.vh = len(self.value_head_names)
        self.head_bias = head_bias
        self.tvf_fixed_head_horizons = None if tvf_fixed_head_horizons is None else list(tvf_fixed_head_horizons)
        self.K = 0 if tvf_fixed_head_horizons is None else len(tvf_fixed_head_horizons)
        # fused head columns
        self.col_policy, self.col_value = 0, n_actions
        self.col_advantage = n_actions + self.vh
        self.col_tvf = 2 * n_actions + self.vh
        self.nh = self.col_tvf + self.K * self.vh
        self._build_parameters(head_scale)
        self.tvf_feature_sparsity, self.tvf_feature_window = tvf_feature_sparsity, tvf_feature_window
        self.tvf_features_mask = None
        self._build_tvf_feature_mask()
        self._bufs: Dict[tuple, torch.Tensor] = {}
        self._rec = None   # launch recorder (see encode)
        # (n_actions, temperature, seed, offset, log_policy, actions, log_pac, raw_policy, values, n_value_heads) of
        # ppo_dense_heads_act_forward_f32, set by the caller of an inference encode(); None again once a launch took it
        self.act_tail = None
        self._chain_split_usable = None  # decided at the first split launch (see _encode_impala)
        self.allow_chain_split = False   # set by a caller that checks chain_split_error() afterwards
        self.obs_index = None  # [B] int32 (device): the next TRAINING forward reads observation i at x[obs_index[i]] (see takes_obs_index)
        self.loss_tail = None  # arguments of the PPO loss for the next training forward's dense + heads launch (ppo_minibatch)
        self._tail_ptrs = {}  # stack index -> pointer arrays of the fused residual-block kernel
        self.obs_norm = None  # shared ObsNormalizer (set by TVFModel when observation_normalization is on)
        self.grad_ready_hook = None  # callable(stream), see _backward_impala (data-parallel gradient buckets)
        self._split_ws = {}  # workspaces of the two-workgroups-per-image chain launch
        self._plans = {}   # (tag, batch, dtype) -> recorded inference launch list
        self.use_plans = True  # False: every inference launch goes through _call (bench.py's per-kernel table brackets it)
        self._build_packed_weights()
        self._build_split_bf16(precision)
        self._build_mlp_fused()
        self._adam_step = 0
        self.exp_avg = None
        self.exp_avg_sq = None

    @property
    def use_tvf(self):
        return self.K > 0

    # ------------------------------------------------------------------ parameters
    def _build_parameters(self, head_scale):
        sp = self.spec
        init = init_parameters(sp, self.n_actions, self.vh, head_scale, self.head_bias, self.K)
        # physical order: encoder..., then the head weights contiguous (one [nh, hidden] matrix),
        # then the head biases contiguous, then log_std
        head_names = [n for n in HEAD_NAMES if f"{n}.weight" in init]
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
        self._state_order = list(init.keys())
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

    def _build_split_bf16(self, precision):
        """--precision=low|medium (rl train.py:166-178 lets cuDNN use TF32 there): the residual blocks of the 32-channel
        stacks run as split-bf16 launches (csrc/stack_bf16x3.hip: three bf16 MFMAs per product, float32 accumulation,
        ~16-bit products; forward and backward-data).  `high` - the default of this class, of bench.py and of every
        parity test - is exact float32 everywhere.  Buffers: per such stack the forward and the transposed packing of
        its four block convolutions, refreshed by ONE launch behind the float32 re-pack."""
        self.precision = precision
        self.split_bf16, self._pk16, self._split_jobs, self._split_conv_jobs = False, {}, None, None
        if precision == "high" or self.encoder_kind != "impala" or self.spec.n_block != 2:
            return
        jobs = []
        nbytes = int(self.lib.ppo_impala_stack_tail_bf16x3_packed_bytes())
        for si, (_cin, cout, _h, _w, ho, wo) in enumerate(self.spec.stacks):
            if not self.lib.ppo_impala_stack_tail_bf16x3_supported(cout, ho, wo):
                continue
            for tr, order in ((0, ((0, 0), (0, 1), (1, 0), (1, 1))), (1, ((1, 1), (1, 0), (0, 1), (0, 0)))):
                buf = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
                self._pk16[(si, tr)] = buf
                ws = [self.params[f"encoder.stacks.{si}.blocks.{bi}.conv{ci}.weight"].data_ptr() for bi, ci in order]
                jobs.append(_lib.SplitPackJob((ctypes.c_void_p * 4)(*ws), buf.data_ptr(), cout, tr))
            names = [f"encoder.stacks.{si}.blocks.{bi}.conv{ci}" for bi in range(2) for ci in range(2)]
            self._pk16[(si, "bias")] = (ctypes.c_void_p * 4)(*[self.params[n + ".bias"].data_ptr() for n in names])
        if jobs:
            self._split_jobs = (_lib.SplitPackJob * len(jobs))(*jobs)
            self.split_bf16 = True
            self._packed_dirty = True
        # the stack-first convolutions that are not part of a chain (csrc/conv_bf16x3.hip): forward and backward-data
        cjobs = []
        self._split_conv_jobs = None
        for si, (cin, cout, h, w, _ho, _wo) in enumerate(self.spec.stacks):
            if not (self.split_bf16 and SPLIT_CONV and si > 0 and cin in (16, 32) and cout in (16, 32)):
                continue
            wname = f"encoder.stacks.{si}.firstconv"
            nb = int(self.lib.ppo_conv3x3_bf16x3_packed_bytes(cin, cout))
            for tr, (ci_op, co_op) in ((0, (cin, cout)), (1, (cout, cin))):
                if not self.lib.ppo_conv3x3_bf16x3_supported(ci_op, co_op, h, w):
                    continue
                buf = torch.zeros(nb, dtype=torch.uint8, device=self.device)
                self._pk16[(wname, tr)] = buf
                cjobs.append(_lib.ConvPackJob(self.params[wname + ".weight"].data_ptr(), buf.data_ptr(), cin, cout, tr))
        if cjobs:
            self._split_conv_jobs = (_lib.ConvPackJob * len(cjobs))(*cjobs)

    def _build_mlp_fused(self):
        """The pointer tables of the fused MLP launches (the flat buffers never move)."""
        self.mlp_fused, self._presummed = False, 0
        sp = self.spec
        if not FUSE_MLP or self.encoder_kind != "mlp" or \
                not self.lib.ppo_mlp_supported(sp.in_features, sp.hidden_units, self.nh):
            return
        P, G = self.params, self.grads
        self._mlp_net = _lib.MlpNet(
            w1=_p(P["encoder.fc1.weight"]), b1=_p(P["encoder.fc1.bias"]), w2=_p(P["encoder.fc2.weight"]),
            b2=_p(P["encoder.fc2.bias"]), wh=_p(self.w_heads), bh=_p(self.b_heads), F=sp.in_features, H=sp.hidden_units,
            NH=self.nh, act=1 if self.encoder_activation_fn == "tanh" else 2)
        self._mlp_grads = _lib.MlpGrads(
            dw1=_p(G["encoder.fc1.weight"]), db1=_p(G["encoder.fc1.bias"]), dw2=_p(G["encoder.fc2.weight"]),
            db2=_p(G["encoder.fc2.bias"]), dwh=_p(self.g_w_heads), dbh=_p(self.g_b_heads), dlog_std=_p(G["log_std"]),
            n_log_std=self.n_actions)
        self._mlp_loss = _lib.MlpLoss()
        self._mlp_npart = ctypes.c_int(0)
        self.mlp_fused = True

    def _mlp_train(self, kind, prev_state, index, stats, n_stats, stat_sums, stat_accumulate, **fields):
        """Forward + loss `kind` + backward of one minibatch through the fused launches; the gradients land in
        self.grad, the per-workgroup sums of g^2 in the optimiser's workspace (adam_step picks them up)."""
        B = int(index.shape[0]) if index is not None else int(prev_state.shape[0])
        x = prev_state
        if x.dtype != torch.float32 or not x.is_contiguous():
            raise ValueError("the mlp encoder takes contiguous float32 observations")
        x_indexed = int(x.shape[0]) if (index is not None and x.shape[0] != B) else 0  # rows of the whole batch
        if self.obs_norm is not None:
            if x_indexed:
                raise ValueError("observation normalisation needs the gathered minibatch")
            xn = self._buf("txn", tuple(x.shape))
            self._call("ppo_obs_normalize_f32", _p(x), 0, _p(self.obs_norm.mu), _p(self.obs_norm.std),
                       self.obs_norm.norm_eps, _p(xn), x.shape[0], self.obs_norm.F)
            x = xn
        L = self._mlp_loss
        L.kind, L.stats = kind, _p(stats)
        for k, v in fields.items():
            setattr(L, k, v)
        H = self.hidden_units
        ws = self._buf("mlp_ws", (int(self.lib.ppo_mlp_train_workspace_floats(B, self.spec.in_features, H, self.nh)),))
        partials = self._ws("adam_ws", self.lib.ppo_adam_workspace_bytes())
        self._call("ppo_mlp_train_f32", _p(x), ctypes.addressof(self._mlp_net), ctypes.addressof(self._mlp_grads), _p(index),
                   x_indexed, B, ctypes.addressof(L), _p(ws), None, _p(stat_sums), n_stats, 1 if stat_accumulate else 0,
                   _p(partials), ctypes.addressof(self._mlp_npart))
        self._presummed = self._mlp_npart.value
        return stats

    def _build_tvf_feature_mask(self):
        """rl/models.py:386-421: a static mask [K, hidden] over the TVF head's weights - `tvf_feature_sparsity` zeroes a
        random share of each head's features (the kept ones scaled by sqrt(1 / keep)), `tvf_feature_window` gives head k
        a window of that many features sliding from the first to the last with k (scaled by sqrt(hidden / window)).
        The initial weights are multiplied by the scaled mask; the 0 / 1 mask stays and is re-applied after every
        optimiser step (mask_feature_weights).  The random mask comes from a CPU generator seeded 99, i.e. what the
        reference draws on --device=cpu (on a GPU it seeds that device's generator: another stream, same law)."""
        if not self.use_tvf or (self.tvf_feature_sparsity <= 0 and self.tvf_feature_window <= 0):
            return
        if self.vh != 1:  # the reference's [K, hidden] mask broadcasts against [K * vh, hidden] only then
            raise ValueError("TVF feature masks need a single value head")
        scaled = tvf_feature_mask(self.K, self.hidden_units, self.tvf_feature_sparsity, self.tvf_feature_window)
        w = self.params["tvf_head.weight"]
        w.mul_(scaled.to(self.device))
        self.tvf_features_mask = torch.gt(scaled, 0).to(torch.uint8).to(self.device).contiguous()

    def mask_feature_weights(self):
        """rl/models.py:425-427; the reference calls it before every forward that evaluates the TVF head (:494-497),
        here it follows whatever wrote the weights (optimiser step, load_state_dict): the same weights at every use."""
        if self.tvf_features_mask is not None:
            w = self.params["tvf_head.weight"]
            self._call("ppo_mask_mul_f32", _p(w), _p(self.tvf_features_mask), w.numel())

    @property
    def early_grad_offset(self) -> int:
        """Offset in the flat gradient from which everything (dense layer, heads, log_std) is written by the first
        launches of a backward pass; [0, offset) are the convolution gradients, final only at its end."""
        return self._offsets["encoder.dense.weight"][0] if self.encoder_kind == "impala" else 0

    def n_parameters(self) -> int:
        return sum(int(np.prod(s)) for _, s in self._offsets.values())

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        """Reference key names (policy_net.<these>), tensors are views into the flat buffer."""
        names = ["log_std"] + [n for n in self._state_order if n != "log_std"]
        return OrderedDict((n, self.params[n]) for n in names)

    # ------------------------------------------------------------------ pre-packed convolution weights
    def _build_packed_weights(self):
        """One buffer with, per convolution, the forward kernels' A operand and the backward-data kernels' (flipped,
        transposed) one in per-lane order, plus the job table of the launch that refreshes all of them."""
        self._pk, self._pack_table, self._packed_dirty, self._scatter = {}, None, False, None
        if not PACKED_WEIGHTS or self.encoder_kind != "impala":
            return
        jobs, total = [], 0
        for name, w in self.params.items():
            if not (name.startswith("encoder.stacks.") and name.endswith(".weight")):
                continue
            cout, cin = int(w.shape[0]), int(w.shape[1])
            for transposed in (0, 1):
                if transposed and name == "encoder.stacks.0.firstconv.weight":
                    continue  # nothing back-propagates into the observations
                n = int(self.lib.ppo_conv3x3_packed_floats(cin, cout, transposed))
                jobs.append((name[:-len(".weight")], transposed, cin, cout, total, n))
                total += n
        self._packed = torch.zeros(total, dtype=torch.float32, device=self.device)
        table = (_lib.PackJob * len(jobs))()
        for k, (wname, transposed, cin, cout, off, n) in enumerate(jobs):
            view = self._packed[off:off + n]
            self._pk[(wname, transposed)] = view
            table[k] = _lib.PackJob(_p(self.params[wname + ".weight"]), _p(view), cin, cout, transposed)
        self._pack_table = table
        self._packed_dirty = True

    def _scatter_table(self):
        """[n_conv, 2] int32 (device): for every element of the flat buffer's convolution head the one or two slots of
        the packed buffer that hold it (-1: none - biases, alignment padding), so that the optimiser step can refresh
        the packed operands itself (ppo_adam_step_scatter_f32).  Found by packing a buffer of element numbers once."""
        if self._scatter is None and self._pack_table is not None:
            n_conv = int(self.early_grad_offset)
            keep = self.flat[:n_conv].clone()
            self.flat[:n_conv] = torch.arange(1, n_conv + 1, dtype=torch.float32, device=self.device)  # exact below 2^24
            self._packed_dirty = True
            self._refresh_packed()
            slots = self._packed.cpu().numpy().astype(np.int64)
            self.flat[:n_conv] = keep
            self._packed_dirty = True
            self._refresh_packed()
            pos = np.flatnonzero(slots > 0)
            src = slots[pos] - 1
            order = np.argsort(src, kind="stable")
            src, pos = src[order], pos[order]
            first = np.r_[True, src[1:] != src[:-1]]
            second = ~first
            assert not (second[1:] & second[:-1]).any(), "a weight sits in more than two packed slots"
            table = np.full((n_conv, 2), -1, np.int32)
            table[src[first], 0] = pos[first]
            table[src[second], 1] = pos[second]
            self._scatter = (torch.from_numpy(table).to(self.device), n_conv)
        return self._scatter

    def mark_weights_changed(self):
        """Call after writing convolution weights other than through adam_step / load_state_dict."""
        self._packed_dirty = self._pack_table is not None or getattr(self, "split_bf16", False)

    def _refresh_packed(self):
        if self._packed_dirty:
            self._packed_dirty = False
            rec, self._rec = self._rec, None  # never part of a recorded forward: the refresh is conditional
            try:
                if self._pack_table is not None:
                    self._call("ppo_conv3x3_pack_weights_f32", ctypes.addressof(self._pack_table), len(self._pack_table))
                if self.split_bf16:
                    self._call("ppo_impala_stack_tail_pack_bf16x3_jobs", ctypes.addressof(self._split_jobs), len(self._split_jobs))
                    if self._split_conv_jobs is not None:
                        self._call("ppo_conv3x3_pack_bf16x3_jobs", ctypes.addressof(self._split_conv_jobs), len(self._split_conv_jobs))
            finally:
                self._rec = rec

    def load_state_dict(self, sd, strict=True):
        self.mark_weights_changed()
        missing = [n for n in self.params if n not in sd]
        unexpected = [n for n in sd if n not in self.params]
        if strict and (missing or unexpected):
            raise KeyError(f"state_dict mismatch: missing {missing}, unexpected {unexpected}")
        with torch.no_grad():
            for n, t in sd.items():
                if n in self.params:
                    self.params[n].copy_(torch.as_tensor(t).to(self.device, torch.float32).reshape(self.params[n].shape))
        self.mask_feature_weights()

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
        fn = getattr(self.lib, fn_name)
        if self._rec is not None:
            self._rec.append((fn, fn_name, args))
        rc = fn(*args, _lib.current_stream())
        if rc != 0:
            _lib.check(rc, fn_name)

    def _linear(self, x, k, wname, out, relu_x=0, tag=""):
        """out[B, n] = f(x)[B, k] @ W[n, k]^T + b  (torch.nn.Linear).  `tag` keys the split-K workspace, so
        forwards that run concurrently on different streams do not share it."""
        w = self.params[wname + ".weight"]
        B, n = x.shape[0], w.shape[0]
        ws_bytes = self.lib.ppo_gemm_workspace_bytes(B, n, k)
        ws = self._ws("gemm_ws" + tag, ws_bytes)
        self._call("ppo_gemm_f32", _p(x), k, 1, relu_x, _p(w), 1, k, 0, _p(self.params[wname + ".bias"]), None,
                   _p(out), n, B, n, k, _p(ws), ws_bytes)

    def _linear_backward(self, x, k, wname, dy, dx, relu_x=0, mask=None, acc=0, side=None):
        """dW = dy^T @ f(x), db = colsum(dy), dx = (dy @ W) [* (mask > 0)]  (dx nullable).  With `side` (a stream
        that already waits for dy) the two parameter gradients go there and only dx stays on the current stream."""
        w = self.params[wname + ".weight"]
        B, n = dy.shape[0], w.shape[0]

        def param_grads():
            self._call("ppo_gemm_f32", _p(dy), 1, n, 0, _p(x), k, 1, relu_x, None, None,
                       _p(self.grads[wname + ".weight"]), k, n, k, B, None, 0)
            if not (wname == "encoder.dense" and getattr(self, "_dense_bias_done", False)):  # heads backward made it
                self._call("ppo_colsum_f32", _p(dy), B, n, n, _p(self.grads[wname + ".bias"]), acc)

        if side is None:
            param_grads()
        else:
            with torch.cuda.stream(side):
                param_grads()
        if dx is not None:
            self._call("ppo_gemm_f32", _p(dy), n, 1, 0, _p(w), k, 1, 0, None, _p(mask), _p(dx), k, B, k, n, None, 0)

    # ------------------------------------------------------------------ forward
    def _conv(self, x, mode, wname, out, residual, n, cin, cout, h, w):
        pk = self._pk.get((wname, 0))
        if pk is not None:
            self._call("ppo_conv3x3_forward_packed_f32", _p(x), mode, _p(pk), _p(self.params[wname + ".bias"]),
                       _p(residual), _p(out), n, cin, cout, h, w)
            return
        self._call("ppo_conv3x3_forward_f32", _p(x), mode, _p(self.params[wname + ".weight"]),
                   _p(self.params[wname + ".bias"]), _p(residual), _p(out), n, cin, cout, h, w)

    def encode(self, x: torch.Tensor, train: bool, tag: Optional[str] = None):
        """x: [B, *input_dims] uint8 (scaled /255 on load; impala only) or float32.  Returns the dict of
        saved tensors; 'h' is the encoder output before the encoder activation, 'hact' its tanh when the
        activation is tanh.  `tag` names the scratch-buffer set ("t" training, "i" inference by default);
        forwards that overlap on different streams use different tags."""
        sp = self.spec
        tag = tag or ("t" if train else "i")
        self._refresh_packed()
        if tuple(x.shape[1:]) != sp.input_dims:
            raise ValueError(f"expected input [B, {sp.input_dims}], got {tuple(x.shape)}")
        if x.dtype not in (torch.uint8, torch.float32) or not x.is_contiguous() or x.device != self.device:
            raise ValueError("input must be a contiguous uint8/float32 tensor on the model's device")
        # Inference forwards (the rollout: 2 x 257 per iteration, with the host on the critical path) replay a
        # recorded launch list: every pointer and size of a (tag, batch) forward is fixed — scratch buffers
        # persist, parameters are views of one flat buffer — except the input, which is patched in.  That cuts
        # the per-launch Python work to the ctypes call itself.
        key = (tag, x.shape[0], x.dtype, self.allow_chain_split)
        plan = None if train or not self.use_plans else self._plans.get(key)
        if plan is not None:
            calls, acts, x_slots = plan
            st, xp = _lib.current_stream(), x.data_ptr()
            for k, (fn, fn_name, args) in enumerate(calls):
                if self.act_tail is not None and fn_name == "ppo_dense_heads_forward_f32":
                    # the rollout's action step rides on the dense + heads launch (Runner._policy_step set the tail)
                    fn_name, tail, self.act_tail = "ppo_dense_heads_act_forward_f32", self.act_tail, None
                    rc = self.lib.ppo_dense_heads_act_forward_f32(*args, *tail, st)
                else:
                    rc = fn(xp, *args[1:], st) if k in x_slots else fn(*args, st)
                if rc != 0:
                    _lib.check(rc, fn_name)
            out = dict(acts)
            if self.obs_norm is None:
                for name in plan_input_keys(self.encoder_kind):
                    out[name] = x
            return out
        if not train and self.use_plans:
            self._rec = []
        try:
            x_in = x
            if self.obs_norm is not None:
                # clamp((x - mu) / (std + eps), -5, 5) ahead of the network (rl/models.py:783-784); the net then
                # sees a float32 input, which is also what its backward reads
                x_in = self._buf(tag + "xn", tuple(x.shape))
                self._call("ppo_obs_normalize_f32", _p(x), 1 if x.dtype == torch.uint8 else 0, _p(self.obs_norm.mu),
                           _p(self.obs_norm.std), self.obs_norm.norm_eps, _p(x_in), x.shape[0], self.obs_norm.F)
            acts = (self._encode_mlp(x_in, train, tag) if self.encoder_kind == "mlp"
                    else self._encode_impala(x_in, train, tag))
            if self.encoder_activation_fn == "tanh" and "heads" not in acts:
                h = acts["h"]
                hact = self._buf(tag + "hact", tuple(h.shape))
                self._call("ppo_tanh_forward_f32", _p(h), _p(hact), h.numel())
                acts["hact"] = hact
            if self._rec is not None:
                xp = x.data_ptr()
                x_slots = frozenset(k for k, (_f, _n, a) in enumerate(self._rec) if a and a[0] == xp)
                self._plans[key] = (self._rec, {k: v for k, v in acts.items() if v is not x}, x_slots)
        finally:
            self._rec = None
        return acts

    def _encode_mlp(self, x, train, tag):
        if x.dtype != torch.float32:
            raise ValueError("the mlp encoder takes float32 observations")
        sp, B = self.spec, x.shape[0]
        if self.mlp_fused and not train:
            # the whole net in one launch: heads, and the encoder output before / after its activation
            o = self._buf(f"{tag}heads", (B, self.nh))
            h = self._buf(f"{tag}h", (B, sp.hidden_units))
            hact = self._buf(f"{tag}hact", (B, sp.hidden_units))
            self._call("ppo_mlp_forward_f32", _p(x), ctypes.addressof(self._mlp_net), None, B, _p(o), _p(h), _p(hact))
            return {"x": x, "h": h, "hact": hact, "heads": o}
        z1 = self._buf(f"{tag}z1", (B, sp.hidden_units))
        self._linear(x, sp.in_features, "encoder.fc1", z1, tag=tag)
        a1 = self._buf(f"{tag}a1", (B, sp.hidden_units))
        self._call("ppo_tanh_forward_f32", _p(z1), _p(a1), z1.numel())
        h = self._buf(f"{tag}h", (B, sp.hidden_units))
        self._linear(a1, sp.hidden_units, "encoder.fc2", h, tag=tag)
        return {"x": x, "a1": a1, "h": h}

    def _stack_full_ptrs(self, si, cin, cout, h, w):
        """Host arrays of the five packed-weight / bias pointers (firstconv + the four block convolutions) of stack
        si for the whole-stack kernel, or None when it does not apply."""
        if not (FUSE_STACK_FULL and FUSE_STACK_TAIL) or self.spec.n_block != 2 or cin != cout \
                or not self.lib.ppo_impala_stack_full_supported(cout, h, w) or self.split_bf16:
            return None  # (split mode: the blocks of the 32-channel stacks have their own launches)
        cached = self._tail_ptrs.get(("full", si))
        if cached is None:
            names = [f"encoder.stacks.{si}.firstconv"] + [f"encoder.stacks.{si}.blocks.{bi}.conv{ci}"
                                                          for bi in range(2) for ci in range(2)]
            pks = [self._pk.get((n, 0)) for n in names]
            if any(pk is None for pk in pks):
                return None
            cached = ((ctypes.c_void_p * 5)(*[pk.data_ptr() for pk in pks]),
                      (ctypes.c_void_p * 5)(*[self.params[n + ".bias"].data_ptr() for n in names]))
            self._tail_ptrs[("full", si)] = cached
        return cached

    def _stack_tail_ptrs(self, si, cout, ho, wo, batch=1 << 30):
        """Host arrays of the four packed-weight / bias pointers of stack si's residual blocks for the fused
        kernel, or None when it does not apply.  The arrays are cached: packed buffers and parameter views keep
        their addresses for the life of the net."""
        if not FUSE_STACK_TAIL or self.spec.n_block != 2 or not self.lib.ppo_impala_stack_tail_supported(cout, ho, wo) \
                or (cout == 16 and (not FUSE_STACK16 or batch < FUSE_STACK16_MIN_BATCH)):
            return None
        cached = self._tail_ptrs.get(si)
        if cached is None:
            names = [f"encoder.stacks.{si}.blocks.{bi}.conv{ci}" for bi in range(2) for ci in range(2)]
            pks = [self._pk.get((n, 0)) for n in names]
            if any(pk is None for pk in pks):
                return None
            cached = ((ctypes.c_void_p * 4)(*[pk.data_ptr() for pk in pks]),
                      (ctypes.c_void_p * 4)(*[self.params[n + ".bias"].data_ptr() for n in names]))
            self._tail_ptrs[si] = cached
        return cached

    def _chain_split_ws(self, tag, B, c, h, w):
        """Exchange slots + flags + control words of ppo_impala_stack_chain_split_forward_f32: zeroed ONCE (the flag
        values only ever grow, the kernel advances its own launch counter), one per scratch-buffer set."""
        key = (tag, B, c, h, w)
        ws = self._split_ws.get(key)
        if ws is None:
            n = int(self.lib.ppo_impala_stack_chain_split_workspace_bytes(B, c, h, w))
            ws = torch.zeros(n, dtype=torch.uint8, device=self.device)
            self._split_ws[key] = ws
        return ws

    def chain_split_error(self) -> bool:
        """True if any split launch saw a workgroup whose partner never arrived (word 2 of the control words).
        One device -> host copy however many workspaces there are."""
        if not self._split_ws:
            return False
        words = torch.stack([ws[-16:].view(torch.int32)[2] for ws in self._split_ws.values()])
        return bool(words.any().item())

    def chain_split_armed(self) -> bool:
        """Whether a forward under allow_chain_split may take the split launch (so its caller has to check for errors)."""
        return bool(CHAIN_SPLIT) and self.encoder_kind == "impala" and self._chain_split_usable is not False

    def chain_split_disable(self):
        """After an error: the one-workgroup launch from now on.  Clears the error (and fault-injection) words and
        forgets every recorded launch list that holds the split launch."""
        self._chain_split_usable = False
        for ws in self._split_ws.values():
            ws[-16:].view(torch.int32)[2:4].zero_()
        self._plans = {k: v for k, v in self._plans.items()
                       if all(name != "ppo_impala_stack_chain_split_forward_f32" for _f, name, _a in v[0])}

    def chain_split_inject_fault(self):
        """Tests only: from the next split launch on, the second workgroup of every pair withholds its flags (control
        word 3, csrc/stack_fused.hip), i.e. the first one times out exactly as if its partner had never been dispatched."""
        for ws in self._split_ws.values():
            ws[-16:].view(torch.int32)[3] = 1

    def _encode_impala(self, x, train, tag):
        sp = self.spec
        index = self.obs_index if train else None
        B = x.shape[0] if index is None else int(index.shape[0])
        acts = {"x": x, "in0_index": index}
        cur, cur_mode = x, (IN_U8 if x.dtype == torch.uint8 else IN_NONE)
        pending = None  # (pooled map, pointer arrays, output buffers) of a stack whose blocks the next launch runs
        for si, (cin, cout, h, w, ho, wo) in enumerate(sp.stacks):
            p = self._buf(f"{tag}p{si}", (B, cout, ho, wo))
            idx = self._buf(f"{tag}idx{si}", (B, cout, ho, wo), torch.uint8) if train else None
            wname = f"encoder.stacks.{si}.firstconv"
            full = self._stack_full_ptrs(si, cin, cout, h, w) if cur_mode == IN_NONE else None
            if full is not None:
                names = [f"{tag}a{si}_0", f"{tag}q{si}_0", f"{tag}a{si}_1", f"{tag}q{si}_1"]
                a0, q0, a1, q1 = (self._buf(nm, (B, cout, ho, wo)) for nm in names)
                outs = (_p(p) if train else None, _p(idx), _p(a0) if train else None, _p(q0) if train else None,
                        _p(a1) if train else None, _p(q1), B, cout, h, w)
                if pending is not None and not train and CHAIN_SPLIT and self.allow_chain_split \
                        and self._chain_split_usable is not False and B <= CHAIN_SPLIT_MAX_BATCH and (cout, h, w) == (32, 21, 21):
                    # a rollout group (at most half as many images as CUs): every image on two workgroups that split
                    # the output channels of the 21x21 convolutions and exchange halves (bit-identical, ~0.7 x the time)
                    p_prev, ptrs_prev, _saves = pending
                    pending = None
                    ws = self._chain_split_ws(tag, B, cout, h, w)
                    split_args = (_p(p_prev), ptrs_prev[0], ptrs_prev[1], full[0], full[1], _p(q1), _p(ws), ws.numel(),
                                  B, cout, h, w)
                    chain_args = (_p(p_prev), ptrs_prev[0], ptrs_prev[1], None, None, None, None, full[0], full[1], *outs)
                    if self._chain_split_usable is None:
                        # First use on this device: the exchange goes through the L2 the two workgroups of a pair share,
                        # which rests on workgroups b and b ^ 8 landing on one XCD.  Both forms once on this batch; if
                        # the bits differ (another partition mode, another dispatch order) the one-workgroup form stays.
                        rec, self._rec = self._rec, None
                        try:
                            self._call("ppo_impala_stack_chain_split_forward_f32", *split_args)
                            got = q1.clone()
                            self._call("ppo_impala_stack_chain_forward_f32", *chain_args)
                            self._chain_split_usable = bool(torch.equal(got, q1)) and not self.chain_split_error()
                        finally:
                            self._rec = rec
                        if not self._chain_split_usable:
                            import warnings
                            warnings.warn("the two-workgroups-per-image chain launch does not reproduce the one-workgroup "
                                          "launch on this device; using the latter (PPO_AMD_CHAIN_SPLIT=0 silences this)")
                        if rec is not None:  # the recorded launch list gets the form that was chosen (q1 holds its result)
                            name = "ppo_impala_stack_chain_split_forward_f32" if self._chain_split_usable \
                                else "ppo_impala_stack_chain_forward_f32"
                            rec.append((getattr(self.lib, name), name, split_args if self._chain_split_usable else chain_args))
                    elif self._chain_split_usable:
                        self._call("ppo_impala_stack_chain_split_forward_f32", *split_args)
                elif pending is not None:
                    # the previous stack's blocks run inside the same launch, on its pooled map
                    p_prev, ptrs_prev, (pa0, pq0, pa1, pq1) = pending
                    pending = None
                    self._call("ppo_impala_stack_chain_forward_f32", _p(p_prev), ptrs_prev[0], ptrs_prev[1],
                               _p(pa0) if train else None, _p(pq0) if train else None, _p(pa1) if train else None,
                               _p(pq1) if train else None, full[0], full[1], *outs)
                else:
                    self._call("ppo_impala_stack_full_forward_f32", _p(cur), full[0], full[1], *outs)
                acts[f"in{si}"], acts[f"idx{si}"] = cur, idx
                acts[f"q{si}_0_in"], acts[f"a{si}_0"], acts[f"q{si}_1_in"], acts[f"a{si}_1"] = p, a0, q0, a1
                cur, cur_mode = q1, IN_NONE
                continue
            if self.split_bf16 and (wname, 0) in self._pk16 and cur_mode != IN_U8 and cur.dtype == torch.float32:
                # --precision=low|medium: the stack-first convolution as split-bf16 products, then the max-pool launch
                if SPLIT_CONV_POOL and self.lib.ppo_conv3x3_pool_bf16x3_supported(cin, cout, h, w):
                    self._call("ppo_conv3x3_pool_bf16x3", _p(cur), int(cur_mode == IN_RELU), _p(self._pk16[(wname, 0)]),
                               _p(self.params[wname + ".bias"]), _p(p), _p(idx), B, cin, cout, h, w)
                else:
                    c = self._buf(f"{tag}c{si}", (B, cout, h, w))
                    self._call("ppo_conv3x3_bf16x3", _p(cur), int(cur_mode == IN_RELU), _p(self._pk16[(wname, 0)]),
                               _p(self.params[wname + ".bias"]), _p(c), B, cin, cout, h, w)
                    self._call("ppo_maxpool3x3s2_forward_f32", _p(c), _p(p), _p(idx), B, cout, h, w)
            elif FUSE_POOL_STACKS >> si & 1:
                # stack-first convolution + max-pool, fused: the pre-pool map never reaches HBM
                pk = self._pk.get((wname, 0))
                if si == 0 and index is not None:
                    # the minibatch's rows of the whole batch, through the permutation: no gather launch, no second copy
                    self._call("ppo_conv3x3_pool_forward_packed_indexed_f32", _p(cur), _p(index), cur_mode, _p(pk),
                               _p(self.params[wname + ".bias"]), _p(p), _p(idx), B, cin, cout, h, w)
                else:
                    self._call("ppo_conv3x3_pool_forward_f32" if pk is None else "ppo_conv3x3_pool_forward_packed_f32",
                               _p(cur), cur_mode, _p(self.params[wname + ".weight"] if pk is None else pk),
                               _p(self.params[wname + ".bias"]), _p(p), _p(idx), B, cin, cout, h, w)
            else:
                c = self._buf(f"{tag}c{si}", (B, cout, h, w))
                self._conv(cur, cur_mode, wname, c, None, B, cin, cout, h, w)
                self._call("ppo_maxpool3x3s2_forward_f32", _p(c), _p(p), _p(idx), B, cout, h, w)
            acts[f"in{si}"], acts[f"idx{si}"] = cur, idx
            q = p
            if self.split_bf16 and (si, 0) in self._pk16:
                # --precision=low|medium: this stack's residual blocks as the split-bf16 launch
                names = [f"{tag}a{si}_0", f"{tag}q{si}_0", f"{tag}a{si}_1", f"{tag}q{si}_1"]
                a0, q0, a1, q1 = (self._buf(nm, (B, cout, ho, wo)) for nm in names)
                if train and SPLIT_SIGNS and cout == 16:  # (measured: 82 -> 61 us for the 16-channel backward chain; nothing at 32 channels)
                    # the backward chain's gates as sign maps (1 byte per 4 elements instead of 4 float32 maps read for a sign)
                    sg = [self._buf(f"{tag}sg{si}_{k}", (B, cout // 4, ho, wo), torch.uint8) for k in range(4)]
                    acts[f"signs{si}"] = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in sg])
                    self._call("ppo_impala_stack_tail_forward_signs_bf16x3", _p(p), _p(self._pk16[(si, 0)]), self._pk16[(si, "bias")],
                               _p(a0), _p(q0), _p(a1), _p(q1), acts[f"signs{si}"], B, cout, ho, wo)
                else:
                    self._call("ppo_impala_stack_tail_forward_bf16x3", _p(p), _p(self._pk16[(si, 0)]), self._pk16[(si, "bias")],
                               _p(a0) if train else None, _p(q0) if train else None, _p(a1) if train else None, _p(q1), B, cout,
                               ho, wo)
                acts[f"q{si}_0_in"], acts[f"a{si}_0"], acts[f"q{si}_1_in"], acts[f"a{si}_1"] = p, a0, q0, a1
                cur, cur_mode = q1, IN_NONE
                continue
            tail = self._stack_tail_ptrs(si, cout, ho, wo, B)
            if tail is not None:
                names = [f"{tag}a{si}_0", f"{tag}q{si}_0", f"{tag}a{si}_1", f"{tag}q{si}_1"]
                a0, q0, a1, q1 = (self._buf(nm, (B, cout, ho, wo)) for nm in names)
                nxt = sp.stacks[si + 1] if si + 1 < len(sp.stacks) else None
                if FUSE_STACK_CHAIN and nxt is not None and (nxt[0], nxt[2], nxt[3]) == (cout, ho, wo) \
                        and self._stack_full_ptrs(si + 1, nxt[0], nxt[1], nxt[2], nxt[3]) is not None:
                    # the next stack's launch takes these blocks along (their output stays in LDS on the way)
                    pending = (p, tail, (a0, q0, a1, q1))
                    acts[f"q{si}_0_in"], acts[f"a{si}_0"], acts[f"q{si}_1_in"], acts[f"a{si}_1"] = p, a0, q0, a1
                    cur, cur_mode = q1, IN_NONE
                    continue
                # inference needs only the stack's output; training keeps the maps the backward pass reads
                # (the 16-channel form re-reads q0 for its second block's skip connection: it is written in inference too)
                self._call("ppo_impala_stack_tail_forward_f32", _p(p), tail[0], tail[1], _p(a0) if train else None,
                           _p(q0) if (train or cout == 16) else None, _p(a1) if train else None, _p(q1), B, cout, ho, wo)
                acts[f"q{si}_0_in"], acts[f"a{si}_0"], acts[f"q{si}_1_in"], acts[f"a{si}_1"] = p, a0, q0, a1
                cur, cur_mode = q1, IN_NONE
                continue
            for bi in range(sp.n_block):
                a = self._buf(f"{tag}a{si}_{bi}", (B, cout, ho, wo))
                qn = self._buf(f"{tag}q{si}_{bi}", (B, cout, ho, wo))
                base = f"encoder.stacks.{si}.blocks.{bi}"
                pk0, pk1 = self._pk.get((base + ".conv0", 0)), self._pk.get((base + ".conv1", 0))
                if not train and FUSE_BLOCK and pk0 is not None and pk1 is not None \
                        and self.lib.ppo_conv3x3_block_supported(cout, ho, wo):
                    # inference below the whole-stack kernel's batch: both convolutions of the block in one launch
                    self._call("ppo_conv3x3_block_forward_packed_f32", _p(q), _p(pk0), _p(self.params[base + ".conv0.bias"]),
                               _p(pk1), _p(self.params[base + ".conv1.bias"]), _p(qn), B, cout, ho, wo)
                    q = qn
                    continue
                self._conv(q, IN_RELU, base + ".conv0", a, None, B, cout, cout, ho, wo)
                self._conv(a, IN_RELU, base + ".conv1", qn, q, B, cout, cout, ho, wo)
                acts[f"q{si}_{bi}_in"], acts[f"a{si}_{bi}"] = q, a
                q = qn
            cur, cur_mode = q, IN_NONE
        flat = cur.view(B, sp.flat)
        h = self._buf(f"{tag}h", (B, sp.hidden_units))
        if self.encoder_activation_fn == "relu":
            # dense layer + fused heads in one call (the K-slice reduction of the dense product and the heads share a
            # launch); heads() hands the result out
            o = self._buf(f"{tag}heads", (B, self.nh))
            w = self.params["encoder.dense.weight"]
            ws_bytes = self.lib.ppo_gemm_workspace_bytes(B, sp.hidden_units, sp.flat)
            ws = self._ws("gemm_ws" + tag, ws_bytes)
            args = (_p(flat), 1, _p(w), _p(self.params["encoder.dense.bias"]), _p(self.w_heads), _p(self.b_heads), 1, _p(h),
                    _p(o), B, sp.flat, sp.hidden_units, self.nh, _p(ws), ws_bytes)
            if train and self.loss_tail is not None:
                # with the discrete PPO loss on the finished head row (ppo_minibatch set the tail): no loss launch
                tail, self.loss_tail = self.loss_tail, None
                self._call("ppo_dense_heads_loss_forward_f32", *args, *tail)
            elif self.act_tail is not None and not train:
                # with the rollout's action step (see encode); the recorded launch list keeps the plain form, whose
                # replay adds the tail of its own env step
                tail, self.act_tail = self.act_tail, None
                rec, self._rec = self._rec, None
                try:
                    self._call("ppo_dense_heads_act_forward_f32", *args, *tail)
                finally:
                    self._rec = rec
                if rec is not None:
                    rec.append((self.lib.ppo_dense_heads_forward_f32, "ppo_dense_heads_forward_f32", args))
            else:
                self._call("ppo_dense_heads_forward_f32", *args)
            acts["heads"] = o
        else:
            self._linear(flat, sp.flat, "encoder.dense", h, relu_x=1, tag=tag)
        acts["flat"], acts["h"] = flat, h
        return acts

    def heads(self, acts, tag="i"):
        """[B, nh] = act(h) @ [policy | value | advantage | tvf]^T (+ bias)  (rl/models.py:467-506).
        `acts` is encode()'s dict (or, for relu nets, the pre-activation tensor h itself)."""
        if isinstance(acts, dict):
            if "heads" in acts:  # computed with the dense layer (encode, IMPALA / relu)
                return acts["heads"]
            h = acts.get("hact", acts["h"])
        else:
            h = acts
            if self.encoder_activation_fn == "tanh":
                raise ValueError("pass encode()'s dict: the tanh activation is computed there")
        B = h.shape[0]
        o = self._buf(f"{tag}heads", (B, self.nh))
        relu = 1 if self.encoder_activation_fn == "relu" else 0
        self._call("ppo_gemm_f32", _p(h), self.hidden_units, 1, relu, _p(self.w_heads), 1, self.hidden_units, 0,
                   _p(self.b_heads), None, _p(o), self.nh, B, self.nh, self.hidden_units, None, 0)
        return o

    def forward(self, x, policy_temperature: float = 1.0, train: bool = False, exclude_value=False,
                exclude_policy=False, exclude_tvf=False, include_features=False,
                required_tvf_heads=None) -> Dict[str, torch.Tensor]:
        """Same result keys as the reference's DualHeadNet.forward (rl/models.py:433-508).  Value-like
        entries are views of the fused head row (strided, not copies)."""
        acts = self.encode(x, train)
        o = self.heads(acts, "t" if train else "i")
        B, nA = o.shape[0], self.n_actions
        result = {}
        if include_features:
            result["raw_features"] = acts["h"]
            result["features"] = acts["hact"] if "hact" in acts else torch.relu(acts["h"])
        if not exclude_policy:
            logp = self._buf("log_policy", (B, nA))
            result["raw_policy"] = o[:, :nA]
            if nA > 32:
                pass  # wide gaussian policies have no log_softmax (rl/rollout.py uses raw_policy only)
            elif policy_temperature > 0:
                self._call("ppo_policy_act_f32", _p(o), B, self.nh, nA, float(policy_temperature), None, 0, 0, 1,
                           _p(logp), None, None, None, None, self.vh)
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
        if not exclude_value:
            result["value"] = o[:, self.col_value:self.col_value + self.vh]
            if self.use_tvf and not exclude_tvf:
                tvf = o[:, self.col_tvf:].view(B, self.K, self.vh)
                result["tvf_value"] = tvf if required_tvf_heads is None else tvf[:, required_tvf_heads]
        result["advantage"] = o[:, self.col_advantage:self.col_advantage + nA]
        result["_heads"], result["_acts"] = o, acts
        return result

    # ------------------------------------------------------------------ backward
    def backward(self, acts, dheads: torch.Tensor, accumulate: bool = False):
        """Back-propagate d loss / d heads through heads and encoder into self.grad."""
        sp = self.spec
        B = dheads.shape[0]
        H = sp.hidden_units
        if accumulate:
            raise NotImplementedError("micro-batch gradient accumulation is not built: with 288 GB of HBM a whole "
                                      "minibatch is one pass (rl/rollout.py:2331-2374 splits only to fit memory)")
        h = acts["h"]
        relu = self.encoder_activation_fn == "relu"
        hin = h if relu else acts["hact"]
        # heads: dW = dheads^T @ act(h); db = colsum(dheads); dh = (dheads @ W) * act'(h).  The parameter gradients
        # (small, latency-bound launches) go to the side stream of the convolution weight gradients; the main stream
        # carries only the dX chain.
        main = torch.cuda.current_stream()
        side = self._wgrad_side_stream() if (WGRAD_SIDE_STREAM and self.encoder_kind == "impala") else None

        dh = self._buf("dh", (B, H))
        # one launch: dh, the heads' weight / bias gradients and - when dh is final (relu) - the dense layer's bias
        # gradient, which is the column sum of dh
        self._dense_bias_done = bool(relu and self.encoder_kind == "impala" and self.nh <= 16)
        if self.nh <= 16:
            self._call("ppo_heads_backward_f32", _p(dheads), _p(hin), 1 if relu else 0, _p(h) if relu else None, _p(self.w_heads),
                       _p(dh), _p(self.g_w_heads), _p(self.g_b_heads) if self.head_bias else None,
                       _p(self.grads["encoder.dense.bias"]) if self._dense_bias_done else None, B, H, self.nh)
        else:
            def head_param_grads():
                self._call("ppo_gemm_f32", _p(dheads), 1, self.nh, 0, _p(hin), H, 1, 1 if relu else 0, None, None,
                           _p(self.g_w_heads), H, self.nh, H, B, None, 0)
                if self.head_bias:
                    self._call("ppo_colsum_f32", _p(dheads), B, self.nh, self.nh, _p(self.g_b_heads), 0)

            if side is None:
                head_param_grads()
            else:
                side.wait_stream(main)  # dheads (and the forward activations) are complete
                with torch.cuda.stream(side):
                    head_param_grads()
            self._call("ppo_gemm_f32", _p(dheads), self.nh, 1, 0, _p(self.w_heads), H, 1, 0, None, _p(h) if relu else None,
                       _p(dh), H, B, H, self.nh, None, 0)
        if not relu:
            self._call("ppo_tanh_backward_f32", _p(dh), _p(hin), _p(dh), dh.numel())
        if self.encoder_kind == "mlp":
            self._backward_mlp(acts, dh)
        else:
            self._backward_impala(acts, dh)

    def _backward_mlp(self, acts, dh):
        sp = self.spec
        B, H = dh.shape
        da1 = self._buf("da1", (B, H))
        self._linear_backward(acts["a1"], H, "encoder.fc2", dh, da1)
        self._call("ppo_tanh_backward_f32", _p(da1), _p(acts["a1"]), _p(da1), da1.numel())
        self._linear_backward(acts["x"], sp.in_features, "encoder.fc1", da1, None)

    def _backward_impala(self, acts, dh):
        """Backward through the encoder.  The weight gradients run on a second stream: a layer's wgrad and its
        backward-data only share inputs, so the wgrad's ramp-up / tail overlaps the next backward-data kernel
        instead of leaving the chip half empty (every kernel here is a persistent grid with a few-microsecond
        prologue and a ragged tail).  Every gradient tensor has its own buffer within a pass, so the two
        streams never reuse memory the other still reads; the main stream joins the side stream at the end."""
        sp, lib = self.spec, self.lib
        B, H = dh.shape
        flat = acts["flat"]
        # dense: dW = dh^T @ relu(flat); db = colsum(dh); dflat = (dh @ W) * (flat > 0)
        c_last, h_last, w_last = sp.out_shape
        g = self._buf(f"g{len(sp.stacks) - 1}_top", (B, c_last, h_last, w_last))
        main = torch.cuda.current_stream()
        side = self._wgrad_side_stream() if WGRAD_SIDE_STREAM else None
        if side is not None:
            side.wait_stream(main)  # dh is complete
        self._linear_backward(flat, sp.flat, "encoder.dense", dh, g, relu_x=1, mask=flat, side=side)
        if self.grad_ready_hook is not None:
            # dense + head gradients (the tail of the flat buffer from `early_grad_offset` on) are final once the
            # launches queued so far on this stream have run: a data-parallel reducer ships them under the rest
            self.grad_ready_hook(side if side is not None else main)

        n_wgrad = [0]

        jobs = []  # deferred slab reductions: one launch for all layers at the end of the pass

        def split_wgrad(mode, cin, cout, hh, ww):
            """--precision=medium|low: this layer's weight gradient as split-bf16 products (csrc/wgrad_bf16x3.hip)."""
            return self.split_bf16 and SPLIT_WGRAD and WGRAD_BATCH_REDUCE and mode in (IN_NONE, IN_RELU) \
                and lib.ppo_conv3x3_backward_weight_bf16x3_supported(cin, cout, hh, ww)

        def wgrad(x, mode, dy, wname, n, cin, cout, hh, ww, argmax=None):
            """argmax given: dy is the POOLED gradient and the kernel forms max-pool backward itself."""
            nbytes = lib.ppo_conv3x3_wgrad_workspace_bytes(cin, cout)
            if WGRAD_BATCH_REDUCE:
                ws = self._ws("wgrad_ws_" + wname, nbytes)  # per layer: the slabs live until the batched reduction
                n_slabs = ctypes.c_int(0)
                if argmax is not None and acts.get("in0_index") is not None and wname == "encoder.stacks.0.firstconv":
                    args_ = ("ppo_conv3x3_backward_weight_slabs_pooled_indexed_f32", _p(x), _p(acts["in0_index"]), mode, _p(dy),
                             _p(argmax), _p(ws), nbytes, n, cin, cout, hh, ww, ctypes.addressof(n_slabs))
                elif argmax is not None:
                    args_ = ("ppo_conv3x3_backward_weight_slabs_pooled_f32", _p(x), mode, _p(dy), _p(argmax), _p(ws),
                             nbytes, n, cin, cout, hh, ww, ctypes.addressof(n_slabs))
                elif split_wgrad(mode, cin, cout, hh, ww):
                    keep = ((ctypes.c_void_p * 1)(x.data_ptr()), (ctypes.c_int * 1)(int(mode == IN_RELU)),
                            (ctypes.c_void_p * 1)(dy.data_ptr()), (ctypes.c_void_p * 1)(ws.data_ptr()))
                    args_ = ("ppo_conv3x3_backward_weight_slabs_batch_bf16x3", keep[0], keep[1], keep[2], keep[3], nbytes, 1, n,
                             cin, cout, hh, ww, ctypes.addressof(n_slabs))
                else:
                    args_ = ("ppo_conv3x3_backward_weight_slabs_f32", _p(x), mode, _p(dy), _p(ws), nbytes, n, cin, cout,
                             hh, ww, ctypes.addressof(n_slabs))
            else:
                ws = self._ws("wgrad_ws", nbytes)
                args_ = ("ppo_conv3x3_backward_weight_f32", _p(x), mode, _p(dy), _p(self.grads[wname + ".weight"]),
                         _p(self.grads[wname + ".bias"]), _p(ws), nbytes, n, cin, cout, hh, ww, 0)
            if side is None:
                self._call(*args_)
            else:
                ev = self._wgrad_events[n_wgrad[0]]
                n_wgrad[0] += 1
                ev.record(main)  # dy (and everything before it) is ready
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    self._call(*args_)
            if WGRAD_BATCH_REDUCE:
                jobs.append(_lib.WgradJob(_p(ws), _p(self.grads[wname + ".weight"]), _p(self.grads[wname + ".bias"]),
                                          n_slabs.value, cin, cout, 0))

        carry = []  # [(x, dy, wname, n, c, hh, ww)]: a first convolution waiting for the next batched launch of its geometry

        def flush_carry():
            while carry:
                x, dy, wname, n, c, hh, ww = carry.pop()
                wgrad(x, IN_NONE, dy, wname, n, c, c, hh, ww)

        def wgrad_blocks(problems, n, c, hh, ww):
            """Weight gradients of a stack's block convolutions, problems = [(x, dy, wname)] (all IN_RELU, c -> c)."""
            if not (WGRAD_BATCH_LAUNCH and WGRAD_BATCH_REDUCE) or len(problems) > 4:
                flush_carry()
                for x, dy, wname in problems:
                    wgrad(x, IN_RELU, dy, wname, n, c, c, hh, ww)
                return
            relu = [1] * len(problems)
            if carry and carry[-1][3:] == (n, c, hh, ww):
                x, dy, wname = carry.pop()[:3]
                problems = list(problems) + [(x, dy, wname)]
                relu.append(0)  # the stack-first convolution read its input raw
            flush_carry()
            nbytes = lib.ppo_conv3x3_wgrad_workspace_bytes(c, c)
            wss = [self._ws("wgrad_ws_" + wname, nbytes) for _x, _dy, wname in problems]
            k = len(problems)
            n_slabs = ctypes.c_int(0)
            ins = (ctypes.c_void_p * k)(*[x.data_ptr() for x, _d, _w in problems])
            dys = (ctypes.c_void_p * k)(*[dy.data_ptr() for _x, dy, _w in problems])
            wsp = (ctypes.c_void_p * k)(*[ws.data_ptr() for ws in wss])
            if split_wgrad(IN_RELU, c, c, hh, ww):
                args_ = ("ppo_conv3x3_backward_weight_slabs_batch_bf16x3", ins, (ctypes.c_int * k)(*relu), dys, wsp, nbytes, k, n,
                         c, c, hh, ww, ctypes.addressof(n_slabs))
            elif k > 4:
                args_ = ("ppo_conv3x3_backward_weight_slabs_batch_mixed_f32", ins, (ctypes.c_int * k)(*relu), dys, wsp, nbytes, k, n,
                         c, c, hh, ww, ctypes.addressof(n_slabs))
            else:
                args_ = ("ppo_conv3x3_backward_weight_slabs_batch_f32", ins, IN_RELU, dys, wsp, nbytes, k, n, c, c, hh, ww,
                         ctypes.addressof(n_slabs))
            if side is None:
                self._call(*args_)
            else:
                ev = self._wgrad_events[n_wgrad[0]]
                n_wgrad[0] += 1
                ev.record(main)  # every dy of the stack is ready
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    self._call(*args_)
            for (_x, _dy, wname), ws in zip(problems, wss):
                jobs.append(_lib.WgradJob(_p(ws), _p(self.grads[wname + ".weight"]), _p(self.grads[wname + ".bias"]),
                                          n_slabs.value, c, c, 0))

        for si in reversed(range(len(sp.stacks))):
            cin, cout, hh, ww, ho, wo = sp.stacks[si]
            full_w = self._stack_full_bwd_ptrs(si, cin, cout, hh, ww) if si > 0 else None
            if full_w is not None:
                # blocks + max-pool backward + transposed first convolution of the stack in one launch
                b0, b1 = f"encoder.stacks.{si}.blocks.0", f"encoder.stacks.{si}.blocks.1"
                p_in, a0, q0, a1 = acts[f"q{si}_0_in"], acts[f"a{si}_0"], acts[f"q{si}_1_in"], acts[f"a{si}_1"]
                da1, g1, da0, g0 = (self._buf(nm, (B, cout, ho, wo)) for nm in
                                    (f"g{si}_1_da", f"g{si}_1_in", f"g{si}_0_da", f"g{si}_0_in"))
                dc = self._buf(f"g{si}_dc", (B, cout, hh, ww))
                g_prev = self._buf(f"g{si - 1}_top", (B, cin, hh, ww))
                masks = (ctypes.c_void_p * 4)(a1.data_ptr(), q0.data_ptr(), a0.data_ptr(), p_in.data_ptr())
                self._call("ppo_impala_stack_full_backward_f32", _p(g), full_w, masks, _p(acts[f"idx{si}"]), _p(da1), _p(g1),
                           _p(da0), _p(g0), _p(dc), _p(g_prev), B, cout, hh, ww)
                wgrad_blocks([(a1, g, b1 + ".conv1"), (q0, da1, b1 + ".conv0"), (a0, g1, b0 + ".conv1"),
                              (p_in, da0, b0 + ".conv0")], B, cout, ho, wo)
                wgrad(acts[f"in{si}"], IN_NONE, dc, f"encoder.stacks.{si}.firstconv", B, cin, cout, hh, ww)
                g = g_prev
                continue
            tail_w = self._stack_tail_bwd_ptrs(si, cout, ho, wo)
            split = self.split_bf16 and (si, 1) in self._pk16
            if split:
                tail_w = True  # the gated transposed chain of this stack's blocks as the split-bf16 launch
            if tail_w is not None:
                # the four backward-data convolutions of the stack's blocks in one launch (csrc/stack_fused.hip)
                b0, b1 = f"encoder.stacks.{si}.blocks.0", f"encoder.stacks.{si}.blocks.1"
                p_in, a0, q0, a1 = acts[f"q{si}_0_in"], acts[f"a{si}_0"], acts[f"q{si}_1_in"], acts[f"a{si}_1"]
                da1, g1, da0, g0 = (self._buf(nm, (B, cout, ho, wo)) for nm in
                                    (f"g{si}_1_da", f"g{si}_1_in", f"g{si}_0_da", f"g{si}_0_in"))
                masks = (ctypes.c_void_p * 4)(a1.data_ptr(), q0.data_ptr(), a0.data_ptr(), p_in.data_ptr())
                if split and acts.get(f"signs{si}") is not None:
                    self._call("ppo_impala_stack_tail_backward_signs_bf16x3", _p(g), _p(self._pk16[(si, 1)]), acts[f"signs{si}"],
                               _p(da1), _p(g1), _p(da0), _p(g0), B, cout, ho, wo)
                elif split:
                    self._call("ppo_impala_stack_tail_backward_bf16x3", _p(g), _p(self._pk16[(si, 1)]), masks, _p(da1), _p(g1),
                               _p(da0), _p(g0), B, cout, ho, wo)
                else:
                    self._call("ppo_impala_stack_tail_backward_f32", _p(g), tail_w, masks, _p(da1), _p(g1), _p(da0), _p(g0),
                               B, cout, ho, wo)
                wgrad_blocks([(a1, g, b1 + ".conv1"), (q0, da1, b1 + ".conv0"), (a0, g1, b0 + ".conv1"),
                              (p_in, da0, b0 + ".conv0")], B, cout, ho, wo)
                g = g0
            defer = WGRAD_BATCH_LAUNCH and WGRAD_BATCH_REDUCE and sp.n_block * 2 <= 4
            problems = []
            for bi in (reversed(range(sp.n_block)) if tail_w is None else ()):
                base = f"encoder.stacks.{si}.blocks.{bi}"
                q_in, a = acts[f"q{si}_{bi}_in"], acts[f"a{si}_{bi}"]
                # g = d loss / d (block output);  block: out = q_in + conv1(relu(conv0(relu(q_in))))
                if defer:
                    problems.append((a, g, base + ".conv1"))
                else:
                    wgrad(a, IN_RELU, g, base + ".conv1", B, cout, cout, ho, wo)
                da = self._buf(f"g{si}_{bi}_da", (B, cout, ho, wo))
                self._call(self._bwd_data_fn(base + ".conv1"), _p(g), _p(self._bwd_w(base + ".conv1")), _p(a), None,
                           _p(da), B, cout, cout, ho, wo)
                if defer:
                    problems.append((q_in, da, base + ".conv0"))
                else:
                    wgrad(q_in, IN_RELU, da, base + ".conv0", B, cout, cout, ho, wo)
                gn = self._buf(f"g{si}_{bi}_in", (B, cout, ho, wo))
                self._call(self._bwd_data_fn(base + ".conv0"), _p(da), _p(self._bwd_w(base + ".conv0")), _p(q_in),
                           _p(g), _p(gn), B, cout, cout, ho, wo)
                g = gn
            if problems:
                wgrad_blocks(problems, B, cout, ho, wo)
            # max-pool backward, then the stack's first convolution
            x_in = acts[f"in{si}"]
            mode = IN_NONE if si > 0 else (IN_U8 if x_in.dtype == torch.uint8 else IN_NONE)
            if si == 0 and WGRAD_POOLED_DY and WGRAD_BATCH_REDUCE \
                    and lib.ppo_conv3x3_backward_weight_pooled_supported(cin, cout, hh, ww):
                # nothing else reads this stack's pre-pool gradient: the weight-gradient kernel forms it band by band
                wgrad(x_in, mode, g, f"encoder.stacks.{si}.firstconv", B, cin, cout, hh, ww, argmax=acts[f"idx{si}"])
                continue
            dc = self._buf(f"g{si}_dc", (B, cout, hh, ww))
            self._call("ppo_maxpool3x3s2_backward_f32", _p(g), _p(acts[f"idx{si}"]), _p(dc), B, cout, hh, ww)
            if WGRAD_RIDE and WGRAD_BATCH_LAUNCH and WGRAD_BATCH_REDUCE and side is None and si > 0 and cin == cout \
                    and mode == IN_NONE and sp.n_block == 2 and sp.stacks[si - 1][1] == cin \
                    and tuple(sp.stacks[si - 1][4:6]) == (hh, ww):
                carry.append((x_in, dc, f"encoder.stacks.{si}.firstconv", B, cin, hh, ww))  # rides in the next stack's launch
            else:
                wgrad(x_in, mode, dc, f"encoder.stacks.{si}.firstconv", B, cin, cout, hh, ww)
            if si > 0:
                g = self._buf(f"g{si - 1}_top", (B, cin, hh, ww))
                fc = f"encoder.stacks.{si}.firstconv"
                if self.split_bf16 and (fc, 1) in self._pk16:
                    self._call("ppo_conv3x3_bf16x3", _p(dc), 0, _p(self._pk16[(fc, 1)]), None, _p(g), B, cout, cin, hh, ww)
                else:
                    self._call(self._bwd_data_fn(fc), _p(dc), _p(self._bwd_w(fc)),
                               None, None, _p(g), B, cin, cout, hh, ww)
        flush_carry()
        if jobs:
            table = (_lib.WgradJob * len(jobs))(*jobs)
            if side is None:
                self._call("ppo_conv3x3_wgrad_reduce_f32", ctypes.addressof(table), len(jobs))
            else:
                with torch.cuda.stream(side):
                    self._call("ppo_conv3x3_wgrad_reduce_f32", ctypes.addressof(table), len(jobs))
        if side is not None:
            main.wait_stream(side)  # all weight gradients are in self.grad before anything reads it

    def _stack_full_bwd_ptrs(self, si, cin, cout, h, w):
        """Host array of the five backward-data packed weights of stack si in processing order (block1.conv1,
        block1.conv0, block0.conv1, block0.conv0, firstconv), or None when the whole-stack kernel does not apply."""
        if not (FUSE_STACK_FULL_BWD and FUSE_STACK_TAIL) or self.spec.n_block != 2 or cin != cout \
                or not self.lib.ppo_impala_stack_full_supported(cout, h, w):
            return None
        cached = self._tail_ptrs.get(("full_bwd", si))
        if cached is None:
            names = [f"encoder.stacks.{si}.blocks.{bi}.conv{ci}" for bi in (1, 0) for ci in (1, 0)]
            names.append(f"encoder.stacks.{si}.firstconv")
            pks = [self._pk.get((n, 1)) for n in names]
            if any(pk is None for pk in pks):
                return None
            cached = (ctypes.c_void_p * 5)(*[pk.data_ptr() for pk in pks])
            self._tail_ptrs[("full_bwd", si)] = cached
        return cached

    def _stack_tail_bwd_ptrs(self, si, cout, ho, wo):
        """Host array of the four backward-data packed weights of stack si's blocks in processing order
        (block1.conv1, block1.conv0, block0.conv1, block0.conv0), or None when the fused kernel does not apply."""
        if not (FUSE_STACK_TAIL and FUSE_STACK_TAIL_BWD >> si & 1) or self.spec.n_block != 2 \
                or not self.lib.ppo_impala_stack_tail_supported(cout, ho, wo) or (cout == 16 and not FUSE_STACK16):
            return None
        cached = self._tail_ptrs.get(("bwd", si))
        if cached is None:
            names = [f"encoder.stacks.{si}.blocks.{bi}.conv{ci}" for bi in (1, 0) for ci in (1, 0)]
            pks = [self._pk.get((n, 1)) for n in names]
            if any(pk is None for pk in pks):
                return None
            cached = (ctypes.c_void_p * 4)(*[pk.data_ptr() for pk in pks])
            self._tail_ptrs[("bwd", si)] = cached
        return cached

    def _bwd_data_fn(self, wname):
        return "ppo_conv3x3_backward_data_packed_f32" if (wname, 1) in self._pk else "ppo_conv3x3_backward_data_f32"

    def _bwd_w(self, wname):
        return self._pk.get((wname, 1), self.params[wname + ".weight"])

    def _wgrad_side_stream(self):
        if getattr(self, "_wgrad_stream", None) is None:
            self._wgrad_stream = torch.cuda.Stream(device=self.device)
            n = len(self.spec.stacks) * (1 + 2 * self.spec.n_block)
            self._wgrad_events = [torch.cuda.Event() for _ in range(n)]
        return self._wgrad_stream

    # ------------------------------------------------------------------ minibatch losses + optimiser
    def takes_obs_index(self, obs) -> bool:
        """Whether a training minibatch can read its observations out of the whole batch through the permutation
        (ppo_conv3x3_pool_forward_packed_indexed_f32 + the indexed first-layer weight gradient) instead of from a
        gathered copy: uint8 images into the fused first convolution + max-pool, the pooled-gradient weight-gradient form."""
        if self.encoder_kind != "impala" or obs.dtype != torch.uint8 or self.obs_norm is not None or not GATHER_IN_CONV:
            return False
        cin, cout, h, w, _ho, _wo = self.spec.stacks[0]
        return bool(FUSE_POOL_STACKS & 1) and self._pk.get(("encoder.stacks.0.firstconv", 0)) is not None \
            and bool(WGRAD_POOLED_DY and WGRAD_BATCH_REDUCE) \
            and bool(self.lib.ppo_conv3x3_backward_weight_pooled_supported(cin, cout, h, w))

    def _train_forward(self, prev_state, index=None):
        if index is not None and prev_state.shape[0] != index.shape[0]:
            if not self.takes_obs_index(prev_state):
                raise ValueError("this net needs the gathered minibatch of observations (takes_obs_index is False)")
            self.obs_index = index
        try:
            acts = self.encode(prev_state, train=True)
        finally:
            self.obs_index = None
        o = self.heads(acts, "t")
        B = o.shape[0]
        return acts, o, B, self._buf("dheads", (B, self.nh))

    def ppo_minibatch(self, prev_state, actions, old_log_pac, old_log_policy, advantages, returns,
                      eps_clip=0.2, ent_coef=0.01, vf_coef=0.5, loss_scale=1.0, index=None, stat_sums=None,
                      stat_accumulate=False):
        """Forward, fused PPO loss, backward: gradients of mean(-gain)*loss_scale land in self.grad
        (Runner.train_policy_minibatch, rl/rollout.py:1610-1771; discrete actions).  vf_coef = 0 (and
        returns None) leaves the value head out, as the dual architecture's policy phase does (:1744).
        prev_state is the (already gathered) minibatch of observations; the per-sample arrays are
        either minibatch-sized or, with ``index`` ([B] int32), whole-batch arrays read at index[b].
        Returns the per-sample statistics tensor [B, 8] (device).  MLP nets on the fused path (mlp_fused) also take
        the whole batch of observations as prev_state (rows read through ``index``) and ``stat_sums``, a device row
        that receives the column sums of the statistics."""
        if self.mlp_fused:
            B = int(index.shape[0]) if index is not None else int(prev_state.shape[0])
            return self._mlp_train(
                _lib.MLP_LOSS_PPO, prev_state, index, self._buf("loss_stats", (B, 8)), 8, stat_sums, stat_accumulate,
                grad_scale=float(loss_scale) / B, n_actions=self.n_actions, n_value_heads=self.vh if returns is not None else 0,
                returns=_p(returns), vf_coef=float(vf_coef), actions_i=_p(actions), old_log_pac=_p(old_log_pac),
                old_log_policy=_p(old_log_policy), advantages=_p(advantages), eps_clip=float(eps_clip), ent_coef=float(ent_coef))
        B = int(index.shape[0]) if index is not None else int(prev_state.shape[0])
        stats = self._buf("loss_stats", (B, 8))
        vh = self.vh if returns is not None else 0
        loss_args = (self.n_actions, vh, _p(actions), _p(old_log_pac), _p(old_log_policy), _p(advantages), _p(returns),
                     float(eps_clip), float(ent_coef), float(vf_coef), float(loss_scale) / B,
                     _p(self._buf("dheads", (B, self.nh))), _p(stats), _p(index))
        # the loss rides on the dense + heads launch of the training forward where that launch exists (IMPALA, relu)
        self.loss_tail = loss_args if FUSE_LOSS else None
        try:
            acts, o, B, dheads = self._train_forward(prev_state, index)
            fused = FUSE_LOSS and self.loss_tail is None
        finally:
            self.loss_tail = None
        if not fused:
            self._call("ppo_ppo_loss_f32", _p(o), B, self.nh, *loss_args)
        self.backward(acts, dheads)
        return stats

    def gaussian_minibatch(self, prev_state, actions, old_log_pac, advantages, returns, eps_clip=0.2, vf_coef=0.5,
                           loss_scale=1.0, index=None, stat_sums=None, stat_accumulate=False):
        """As ppo_minibatch for gaussian policies (rl/rollout.py:1693-1704); also fills log_std's gradient."""
        if self.mlp_fused:
            B = int(index.shape[0]) if index is not None else int(prev_state.shape[0])
            return self._mlp_train(
                _lib.MLP_LOSS_GAUSSIAN, prev_state, index, self._buf("loss_stats", (B, 8)), 8, stat_sums, stat_accumulate,
                grad_scale=float(loss_scale) / B, n_actions=self.n_actions, n_value_heads=self.vh if returns is not None else 0,
                returns=_p(returns), vf_coef=float(vf_coef), actions_f=_p(actions), old_log_pac=_p(old_log_pac),
                advantages=_p(advantages), log_std=_p(self.params["log_std"]), eps_clip=float(eps_clip),
                dlog_std_rows=_p(self._buf("dlog_std_rows", (B, self.n_actions))))
        acts, o, B, dheads = self._train_forward(prev_state, index)
        stats = self._buf("loss_stats", (B, 8))
        rows = self._buf("dlog_std_rows", (B, self.n_actions))
        vh = self.vh if returns is not None else 0
        self._call("ppo_gaussian_loss_f32", _p(o), B, self.nh, self.n_actions, vh, _p(actions), _p(old_log_pac),
                   _p(advantages), _p(returns), _p(self.params["log_std"]), float(eps_clip), float(vf_coef),
                   float(loss_scale) / B, _p(dheads), _p(rows), _p(stats), _p(index))
        # log_std's gradient first: it sits in the early data-parallel bucket, which leaves during the backward pass
        self._call("ppo_colsum_f32", _p(rows), B, self.n_actions, self.n_actions, _p(self.grads["log_std"]), 0)
        self.backward(acts, dheads)
        return stats

    def value_minibatch(self, prev_state, returns=None, tvf_returns=None, tvf_weights=None, vf_coef=0.5,
                        tvf_coef=1.0, loss_scale=1.0, index=None, tvf_keep_prob=1.0, dropout_seed=0, dropout_offset=0,
                        stat_sums=None, stat_accumulate=False):
        """Value phase (Runner.train_value_minibatch, rl/rollout.py:1513-1567; TVF loss rl/tvf.py:32-77, with
        horizon dropout when tvf_keep_prob < 1: :64-69)."""
        if self.mlp_fused:
            B = int(index.shape[0]) if index is not None else int(prev_state.shape[0])
            return self._mlp_train(
                _lib.MLP_LOSS_VALUE, prev_state, index, self._buf("value_stats", (B, 4)), 4, stat_sums, stat_accumulate,
                grad_scale=float(loss_scale) / B, value_col=self.col_value, n_value_heads=self.vh if returns is not None else 0,
                returns=_p(returns), vf_coef=float(vf_coef), tvf_col=self.col_tvf if self.K else 0,
                n_tvf=self.K if tvf_returns is not None else 0, tvf_stride=max(self.vh, 1), tvf_returns=_p(tvf_returns),
                tvf_weights=_p(tvf_weights), tvf_coef=float(tvf_coef), tvf_keep_prob=float(tvf_keep_prob),
                seed=int(dropout_seed) & (2**64 - 1), offset=int(dropout_offset) & (2**64 - 1))
        acts, o, B, dheads = self._train_forward(prev_state, index)
        stats = self._buf("value_stats", (B, 4))
        self._call("ppo_value_loss_f32", _p(o), B, self.nh, self.col_value, self.vh if returns is not None else 0,
                   _p(returns), float(vf_coef), self.col_tvf if self.K else 0, self.K if tvf_returns is not None else 0,
                   max(self.vh, 1), _p(tvf_returns), _p(tvf_weights), float(tvf_coef), float(loss_scale) / B, _p(dheads),
                   _p(stats), _p(index), float(tvf_keep_prob), int(dropout_seed) & (2**64 - 1),
                   int(dropout_offset) & (2**64 - 1))
        self.backward(acts, dheads)
        return stats

    def distil_minibatch(self, prev_state, targets, old_policy, beta=1.0, use_tvf=False, weights=None, gaussian=False,
                         loss_scale=1.0, index=None, stat_sums=None, stat_accumulate=False):
        """Distillation phase (Runner.train_distil_minibatch, rl/rollout.py:1331-1449): targets [*] against the
        ext value head, or [*, K] against the TVF heads' ext column (use_tvf)."""
        col, n_pred, stride = (self.col_tvf, self.K, self.vh) if use_tvf else (self.col_value, 1, 1)
        if self.mlp_fused:
            B = int(index.shape[0]) if index is not None else int(prev_state.shape[0])
            return self._mlp_train(
                _lib.MLP_LOSS_DISTIL, prev_state, index, self._buf("distil_stats", (B, 4)), 4, stat_sums, stat_accumulate,
                grad_scale=float(loss_scale) / B, n_actions=self.n_actions, pred_col=col, n_pred=n_pred, pred_stride=stride,
                vector_targets=1 if use_tvf else 0, targets=_p(targets), weights=_p(weights), old_policy=_p(old_policy),
                log_std=_p(self.params["log_std"]) if gaussian else None, beta=float(beta))
        acts, o, B, dheads = self._train_forward(prev_state, index)
        stats = self._buf("distil_stats", (B, 4))
        self._call("ppo_distil_loss_f32", _p(o), B, self.nh, self.n_actions, col, n_pred, stride, 1 if use_tvf else 0,
                   _p(targets), _p(weights), _p(old_policy), _p(self.params["log_std"]) if gaussian else None,
                   float(beta), float(loss_scale) / B, _p(dheads), _p(stats), _p(index))
        self.backward(acts, dheads)
        return stats

    def last_dheads(self, B):
        """d loss / d heads [B, nh] of the last training minibatch (a view of scratch memory, for tests and diagnostics):
        the fused MLP path keeps it at the end of its workspace, the op-by-op path in its own buffer."""
        if self.mlp_fused:
            ws = self._buf("mlp_ws", (int(self.lib.ppo_mlp_train_workspace_floats(B, self.spec.in_features, self.hidden_units,
                                                                                   self.nh)),))
            return ws[4 * B * self.hidden_units:4 * B * self.hidden_units + B * self.nh].view(B, self.nh)
        return self._buf("dheads", (B, self.nh))

    def zero_untouched_grads(self):
        """log_std only receives gradient from the gaussian policy loss; every other parameter is overwritten
        by each backward.  Called by phases that do not touch log_std so a stale gradient never leaks."""
        self.grads["log_std"].zero_()

    def adam_step(self, lr=2.5e-4, beta1=0.9, beta2=0.999, eps=1e-5, max_grad_norm=20.0, grad_div=1.0,
                  grad_norm_out: Optional[torch.Tensor] = None, state: Optional["AdamState"] = None):
        """clip_grad_norm_ + torch.optim.Adam.step over the flat buffer (rl/rollout.py:1287-1321).
        `state`: a separate set of Adam moments over the same parameters (the reference's distil optimiser,
        rl/rollout.py:136-141); default is the net's own."""
        ws = self._ws("adam_ws", self.lib.ppo_adam_workspace_bytes())
        scatter = self._scatter_table() if ADAM_SCATTER else None
        if scatter is None:
            self.mark_weights_changed()
        elif self._packed_dirty:
            self._refresh_packed()  # weights written by hand since the last step: start from a consistent packed copy
        if state is not None:
            state.ensure(self.flat)
            state.step += 1
            m, v, step = state.exp_avg, state.exp_avg_sq, state.step
        else:
            if self.exp_avg is None:
                self.exp_avg = torch.zeros_like(self.flat)
                self.exp_avg_sq = torch.zeros_like(self.flat)
            self._adam_step += 1
            m, v, step = self.exp_avg, self.exp_avg_sq, self._adam_step
        n_part, self._presummed = self._presummed, 0
        if n_part and scatter is None and grad_div == 1.0:
            # the launch that wrote the gradients left the per-workgroup sums of g^2 in the workspace (fused MLP path)
            self._call("ppo_adam_step_presummed_f32", _p(self.flat), _p(self.grad), _p(m), _p(v), self.flat.numel(), step,
                       float(lr), float(beta1), float(beta2), float(eps), float(max_grad_norm), float(grad_div), _p(ws),
                       n_part, _p(grad_norm_out))
            self.mask_feature_weights()
            return
        if scatter is not None:
            # the step also refreshes the convolution kernels' packed operands: no re-pack launch before the next forward
            self._call("ppo_adam_step_scatter_f32", _p(self.flat), _p(self.grad), _p(m), _p(v), self.flat.numel(), step,
                       float(lr), float(beta1), float(beta2), float(eps), float(max_grad_norm), float(grad_div), _p(ws),
                       _p(grad_norm_out), _p(scatter[0]), scatter[1], _p(self._packed))
            self.mask_feature_weights()
            return
        self._call("ppo_adam_step_f32", _p(self.flat), _p(self.grad), _p(m), _p(v), self.flat.numel(), step, float(lr),
                   float(beta1), float(beta2), float(eps), float(max_grad_norm), float(grad_div), _p(ws),
                   _p(grad_norm_out))
        self.mask_feature_weights()

    # ------------------------------------------------------------------ optimiser state (checkpoints)
    def parameter_order(self):
        """Parameter names in the order of the reference module's `.parameters()` (= its state_dict order): the
        integer keys of torch.optim.Adam.state_dict()['state'] index this list."""
        return list(self.state_dict().keys())

    def adam_state_dict(self, exp_avg, exp_avg_sq, step, cfg=None):
        """The layout of `torch.optim.Adam(net.parameters()).state_dict()` — what the reference stores per optimiser
        (rl/rollout.py:412-421): {'state': {i: {'step': f32 scalar tensor, 'exp_avg', 'exp_avg_sq'}}, 'param_groups':
        [{'lr', 'betas', 'eps', 'weight_decay', 'amsgrad', 'maximize', 'foreach', 'capturable', 'params': [0..n)}]},
        i = index into `parameter_order()`.  torch creates a parameter's entry at the first step that sees a gradient
        for it, so parameters whose gradient was never non-zero (the reference leaves them at grad=None) have none."""
        names = self.parameter_order()
        state = {}
        if exp_avg is not None and step > 0:
            touched = {}
            for i, name in enumerate(names):
                o, shape = self._offsets[name]
                n = int(np.prod(shape))
                touched[i] = (exp_avg_sq[o:o + n], exp_avg[o:o + n], shape)
            flags = torch.stack([(v != 0).any() | (m != 0).any() for v, m, _ in touched.values()]).cpu().tolist()
            for (i, (v, m, shape)), hit in zip(touched.items(), flags):
                if hit:
                    state[i] = {"step": torch.tensor(float(step), dtype=torch.float32),
                                "exp_avg": m.view(shape).clone(), "exp_avg_sq": v.view(shape).clone()}
        group = {"lr": float(cfg.lr) if cfg is not None else 0.0,
                 "betas": (float(cfg.adam_beta1), float(cfg.adam_beta2)) if cfg is not None else (0.9, 0.999),
                 "eps": float(cfg.adam_epsilon) if cfg is not None else 1e-8, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def read_adam_state_dict(self, sd):
        """(exp_avg, exp_avg_sq, step) flat buffers from a torch.optim.Adam-layout dict (see adam_state_dict) or from
        the two layouts earlier versions of this package wrote (name-keyed per-parameter moments; whole flat buffers).
        (None, None, 0) when the optimiser had not stepped."""
        if "flat" in sd:  # round <= 2: separate flat state of the distil optimiser
            return (sd["flat"]["exp_avg"].to(self.device).clone(), sd["flat"]["exp_avg_sq"].to(self.device).clone(),
                    int(sd["step"]))
        state = sd.get("state") or {}
        if not state:
            return None, None, int(sd.get("step", 0)) if not isinstance(sd.get("step"), torch.Tensor) else 0
        names = self.parameter_order()
        m, v, step = torch.zeros_like(self.flat), torch.zeros_like(self.flat), int(sd.get("step", 0) or 0)
        for key, st in state.items():
            name = names[int(key)] if not isinstance(key, str) or key.isdigit() else key
            o, shape = self._offsets[name]
            n = int(np.prod(shape))
            m[o:o + n].copy_(torch.as_tensor(st["exp_avg"]).reshape(-1))
            v[o:o + n].copy_(torch.as_tensor(st["exp_avg_sq"]).reshape(-1))
            if "step" in st:
                step = max(step, int(float(st["step"])))
        return m, v, step

    def optimizer_state_dict(self, cfg=None):
        return self.adam_state_dict(self.exp_avg, self.exp_avg_sq, self._adam_step, cfg)

    def load_optimizer_state_dict(self, sd):
        self.exp_avg, self.exp_avg_sq, self._adam_step = self.read_adam_state_dict(sd)


class TVFModel:
    """Host mirror of the reference's TVFModel (rl/models.py:511-856): owns `policy_net` and `value_net`
    (the same object for `architecture='single'`, two nets for 'dual'), exposes
    `forward(x, output=..., policy_temperature=...) -> dict` with the reference's routing and key aliasing
    (:790-821) and a `state_dict` with the reference's `policy_net.` / `value_net.` prefixes."""

    def __init__(self, encoder: str, encoder_args=None, input_dims=(4, 84, 84), actions: int = 6, device="cuda",
                 architecture: str = "dual", dtype=torch.float32, use_rnd: bool = False, hidden_units: int = 512,
                 encoder_activation_fn: str = "relu", observation_normalization=False,
                 freeze_observation_normalization=False, tvf_fixed_head_horizons=None, tvf_fixed_head_weights=None,
                 tvf_feature_sparsity: float = 0.0, tvf_feature_window: int = -1, head_scale: float = 1.0,
                 value_head_names=("ext",), norm_eps: float = 1e-5, head_bias: bool = False,
                 observation_scaling: str = "scaled", precision: str = "high"):
        if architecture not in ("single", "dual"):
            raise Exception("Invalid architecture, use [dual|single]")
        if use_rnd:
            raise NotImplementedError("RND is outside the PPO hot path (DESIGN.md)")
        if dtype != torch.float32:
            raise ValueError("the reference path is float32 (rl/models.py:31-32)")
        if observation_scaling != "scaled":
            raise NotImplementedError("observation_scaling='scaled' only (x/255 fused into the first conv)")
        if isinstance(encoder_args, str):
            import ast
            encoder_args = ast.literal_eval(encoder_args)
        self.input_dims = tuple(input_dims)
        self.actions = actions
        self.dtype = dtype
        self.architecture = architecture
        self.encoder_name = encoder
        self.tvf_fixed_head_horizons = tvf_fixed_head_horizons
        self.tvf_fixed_head_weights = tvf_fixed_head_weights
        if architecture == "single":
            self.name = "PPO-" + encoder
        else:
            self.name = ("TVF-" if tvf_fixed_head_horizons is not None else "DNA-") + encoder

        def make_net():
            return DualHeadNet(encoder, input_dims, actions, hidden_units=hidden_units,
                               activation_fn=encoder_activation_fn, tvf_fixed_head_horizons=tvf_fixed_head_horizons,
                               tvf_feature_sparsity=tvf_feature_sparsity, tvf_feature_window=tvf_feature_window,
                               head_scale=head_scale, value_head_names=value_head_names, head_bias=head_bias,
                               device=device, precision=precision, **(encoder_args or {}))

        self.policy_net = make_net()
        self.value_net = make_net() if architecture == "dual" else self.policy_net
        self.device = self.policy_net.device
        self.observation_normalization = bool(observation_normalization)
        self.obs_norm = None
        if self.observation_normalization:
            self.obs_norm = ObsNormalizer(self.input_dims, self.device, norm_eps=norm_eps,
                                          frozen=freeze_observation_normalization)
            self.policy_net.obs_norm = self.value_net.obs_norm = self.obs_norm

    def model_size(self, trainable_only: bool = True):
        n = self.policy_net.n_parameters()
        return n if self.architecture == "single" else n + self.value_net.n_parameters()

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
        x = self.prep_for_model(x)
        if update_normalization and self.obs_norm is not None:
            self.obs_norm.update(x)
        net_args = dict(policy_temperature=policy_temperature, include_features=include_features, **kwargs)

        def public(d):
            return {k: (v.clone() if self.architecture == "dual" else v) for k, v in d.items() if not k.startswith("_")}

        result = {}
        if self.architecture == "single":
            for k, v in public(self.policy_net.forward(x, **net_args)).items():
                result["policy_" + k] = v
                result["value_" + k] = v
                result[k] = v
            return result
        # dual: head rows live in per-net scratch that the next forward of that net reuses, so results are
        # cloned (small [B, heads] tensors)
        if output == "full":
            for k, v in public(self.policy_net.forward(x, **net_args)).items():
                result["policy_" + k] = v
            for k, v in public(self.value_net.forward(x, **net_args)).items():
                result["value_" + k] = v
            return result
        if output in ("default", "policy"):
            result.update(public(self.policy_net.forward(x, **net_args, exclude_value=output == "default")))
        if output in ("default", "value"):
            result.update(public(self.value_net.forward(x, **net_args, exclude_policy=output == "default")))
        return result

    __call__ = forward

    def log_policy(self, x):
        return self.forward(x, output="policy")["log_policy"].detach().cpu().numpy()

    @property
    def obs_rms(self):
        """The reference's `model.obs_rms` (utils.RunningMeanStd) as host arrays: mean / var [*input_dims] float64
        and the scalar count."""
        import types
        n = self.obs_norm
        return types.SimpleNamespace(mean=n.mean.cpu().numpy().reshape(self.input_dims),
                                     var=n.var.cpu().numpy().reshape(self.input_dims), count=n.count)

    def perform_normalization(self, x, update_normalization: bool = False):
        """rl/models.py:666-694 on a prepared batch; returns the normalised float32 tensor."""
        x = self.prep_for_model(x)
        if update_normalization:
            self.obs_norm.update(x)
        return self.obs_norm.apply(x, torch.empty(x.shape, dtype=torch.float32, device=self.device))

    def adjust_value_scale(self, factor: float, process_value=True, process_tvf=True, value_net_only=False):
        """rl/models.py:630-651: scale value predictions by scaling the value / TVF head weights and biases."""
        nets = [self.value_net] if value_net_only else ([self.policy_net] if self.architecture == "single" else
                                                        [self.policy_net, self.value_net])
        for net in nets:
            names = (["value_head"] if process_value else []) + (["tvf_head"] if process_tvf and net.use_tvf else [])
            for n in names:
                net.params[n + ".weight"].mul_(factor)
                if net.head_bias:
                    net.params[n + ".bias"].mul_(factor)

    def state_dict(self):
        sd = OrderedDict()
        for prefix, net in (("policy_net.", self.policy_net), ("value_net.", self.value_net)):
            for k, v in net.state_dict().items():
                sd[prefix + k] = v
        return sd

    def load_state_dict(self, sd, strict=True):
        pol = {k[len("policy_net."):]: v for k, v in sd.items() if k.startswith("policy_net.")}
        self.policy_net.load_state_dict(pol, strict=strict)
        if self.architecture == "dual":
            val = {k[len("value_net."):]: v for k, v in sd.items() if k.startswith("value_net.")}
            self.value_net.load_state_dict(val, strict=strict)
