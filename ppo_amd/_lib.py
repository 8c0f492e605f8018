"""ctypes binding of libppo_amd.so (the C ABI declared in include/ppo_amd.h).

The library is the product: there is no CPU or PyTorch fallback.  A missing
library, a missing symbol or a missing GPU raises — loudly, at the call site.

torch is imported before the library is opened so that the HIP runtime already
mapped by PyTorch-ROCm (SONAME libamdhip64.so.7) is the one our kernels and
streams live in; tensors' ``data_ptr()`` and ``torch.cuda.current_stream()``
are then directly usable as the plain pointers / hipStream_t of the C ABI.
"""
import ctypes
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PPO_AMD_LIB") or os.path.join(HERE, "lib", "libppo_amd.so")  # override: tuning builds

PPO_TERM_NONE, PPO_TERM_U8, PPO_TERM_F32 = 0, 1, 2
PPO_SCAN_AUTO, PPO_SCAN_COLUMNS, PPO_SCAN_TILES = 0, 1, 2
PPO_IN_NONE, PPO_IN_RELU, PPO_IN_U8 = 0, 1, 2

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_d = ctypes.c_double
_f = ctypes.c_float
_sz = ctypes.c_size_t
_u64 = ctypes.c_uint64

# name -> (restype, argtypes); must list every symbol include/ppo_amd.h declares
# (tests/test_abi.py parses the header and checks this table and the .so against it)
SIGNATURES = {
    "ppo_version": (_i, []),
    "ppo_last_error": (ctypes.c_char_p, []),
    "ppo_gae_scan_f32": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i64, _d, _d, _d, _i, _vp]),
    "ppo_bootstrapped_returns_f32": (_i, [_vp, _vp, _i, _vp, _vp, _d, _vp, _i, _i, _i64, _vp]),
    "ppo_conv3x3_forward_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ppo_conv3x3_backward_data_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ppo_conv3x3_wgrad_workspace_bytes": (_sz, [_i, _i]),
    "ppo_conv3x3_backward_weight_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp]),
    "ppo_maxpool3x3s2_forward_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ppo_maxpool3x3s2_backward_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ppo_gemm_workspace_bytes": (_sz, [_i, _i, _i]),
    "ppo_gemm_f32": (_i, [_vp, _i64, _i64, _i, _vp, _i64, _i64, _i, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _sz, _vp]),
    "ppo_colsum_f32": (_i, [_vp, _i, _i, _i64, _vp, _i, _vp]),
    "ppo_heads_backward_f32": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ppo_dense_heads_forward_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "ppo_dense_heads_act_forward_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _i, _f, _u64,
                                             _u64, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "ppo_dense_heads_loss_forward_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _i, _i, _vp, _vp, _vp,
                                              _vp, _vp, _f, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "ppo_policy_act_f32": (_i, [_vp, _i, _i, _i, _f, _vp, _u64, _u64, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "ppo_gather_rows": (_i, [_vp, _i64, _i64, _vp, _i, _vp, _vp]),
    "ppo_moments_workspace_bytes": (_sz, []),
    "ppo_moments_f64": (_i, [_vp, _i64, _vp, _vp, _vp]),
    "ppo_normalize_f32": (_i, [_vp, _i64, _vp, _f, _vp, _vp, _vp]),
    "ppo_accumulate_f32": (_i, [_vp, _vp, _i64, _vp]),
    "ppo_mask_mul_f32": (_i, [_vp, _vp, _i64, _vp]),
    "ppo_tvf_returns_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ppo_tvf_returns_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _d, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp,
                                 _vp, _sz, _vp, _vp]),
    "ppo_synth_env_create": (_vp, [_i, _i64, _u64, _d, _i64, _i]),
    "ppo_synth_env_destroy": (None, [_vp]),
    "ppo_synth_env_reset": (_i, [_vp, _vp]),
    "ppo_synth_env_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ppo_synth_env_step_upload": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "ppo_synth_env_get_state": (_i, [_vp, _vp, _vp, _vp]),
    "ppo_synth_env_set_state": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "ppo_ppo_loss_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "ppo_conv3x3_pool_forward_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ppo_conv3x3_packed_floats": (_sz, [_i, _i, _i]),
    "ppo_conv3x3_pack_weights_f32": (_i, [_vp, _i, _vp]),
    "ppo_conv3x3_forward_packed_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ppo_conv3x3_backward_data_packed_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ppo_conv3x3_pool_forward_packed_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ppo_conv3x3_pool_forward_packed_indexed_f32": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ppo_conv1_pool_form": (_i, [_i]),
    "ppo_conv3x3_block_supported": (_i, [_i, _i, _i]),
    "ppo_conv3x3_block_forward_packed_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ppo_conv3x3_backward_weight_slabs_f32": (_i, [_vp, _i, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp, _vp]),
    "ppo_conv3x3_backward_weight_pooled_supported": (_i, [_i, _i, _i, _i]),
    "ppo_conv3x3_backward_weight_slabs_pooled_f32": (_i, [_vp, _i, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp, _vp]),
    "ppo_conv3x3_backward_weight_slabs_pooled_indexed_f32": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp, _vp]),
    "ppo_conv3x3_backward_weight_slabs_batch_f32": (_i, [_vp, _i, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "ppo_conv3x3_backward_weight_slabs_batch_mixed_f32": (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "ppo_conv3x3_backward_weight_bf16x3_supported": (_i, [_i, _i, _i, _i]),
    "ppo_conv3x3_backward_weight_slabs_batch_bf16x3": (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "ppo_conv3x3_wgrad_reduce_f32": (_i, [_vp, _i, _vp]),
    "ppo_tanh_forward_f32": (_i, [_vp, _vp, _sz, _vp]),
    "ppo_tanh_backward_f32": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "ppo_value_loss_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _f, _i, _i, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _f, _u64, _u64, _vp]),
    "ppo_distil_loss_f32": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp]),
    "ppo_gaussian_act_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _u64, _u64, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "ppo_gaussian_loss_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "ppo_impala_stack_tail_supported": (_i, [_i, _i, _i]),
    "ppo_impala_stack_tail_forward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ppo_impala_stack_full_supported": (_i, [_i, _i, _i]),
    "ppo_impala_stack_full_backward_f32": (_i, [_vp] * 10 + [_i, _i, _i, _i, _vp]),
    "ppo_impala_stack_chain_forward_f32": (_i, [_vp] * 15 + [_i, _i, _i, _i, _vp]),
    "ppo_impala_stack_chain_split_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ppo_impala_stack_chain_split_forward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _vp]),
    "ppo_impala_stack_full_forward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ppo_impala_stack_tail_backward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ppo_obs_moments_f64": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "ppo_obs_rms_update_f64": (_i, [_vp, _d, _d, _vp, _vp, _vp, _vp, _i, _vp]),
    "ppo_obs_normalize_f32": (_i, [_vp, _i, _vp, _vp, _f, _vp, _i, _i, _vp]),
    "ppo_adam_workspace_bytes": (_sz, []),
    "ppo_adam_step_f32": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _d, _d, _d, _d, _f, _f, _vp, _vp, _vp]),
    "ppo_adam_step_presummed_f32": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _d, _d, _d, _d, _f, _f, _vp, _i, _vp, _vp]),
    "ppo_impala_stack_tail_bf16x3_packed_bytes": (_sz, []),
    "ppo_impala_stack_tail_bf16x3_supported": (_i, [_i, _i, _i]),
    "ppo_impala_stack_tail_pack_bf16x3": (_i, [_vp, _vp, _i, _i, _vp]),
    "ppo_impala_stack_tail_pack_bf16x3_jobs": (_i, [_vp, _i, _vp]),
    "ppo_impala_stack_tail_forward_bf16x3": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ppo_impala_stack_tail_backward_bf16x3": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ppo_impala_stack_tail_bf16x3_sign_bytes": (_sz, [_i, _i, _i, _i]),
    "ppo_impala_stack_tail_forward_signs_bf16x3": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ppo_impala_stack_tail_backward_signs_bf16x3": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ppo_conv3x3_bf16x3_supported": (_i, [_i, _i, _i, _i]),
    "ppo_conv3x3_bf16x3_packed_bytes": (_sz, [_i, _i]),
    "ppo_conv3x3_pack_bf16x3_jobs": (_i, [_vp, _i, _vp]),
    "ppo_conv3x3_bf16x3": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ppo_conv3x3_pool_bf16x3_supported": (_i, [_i, _i, _i, _i]),
    "ppo_conv3x3_backward_weight_slabs_batch_bf16_split": (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "ppo_conv3x3_bf16_split_packed_bytes": (_sz, [_i, _i, _i]),
    "ppo_conv3x3_pack_bf16_split": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "ppo_conv3x3_bf16_split": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ppo_conv3x3_pool_bf16x3": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ppo_mlp_supported": (_i, [_i, _i, _i]),
    "ppo_mlp_forward_f32": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "ppo_mlp_train_workspace_floats": (_sz, [_i, _i, _i, _i]),
    "ppo_mlp_train_f32": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "ppo_adam_step_scatter_f32": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _d, _d, _d, _d, _f, _f, _vp, _vp, _vp, _i64, _vp, _vp]),
}

class PackJob(ctypes.Structure):
    """ppo_pack_job (include/ppo_amd.h)."""
    _fields_ = [("weight", ctypes.c_void_p), ("packed", ctypes.c_void_p), ("cin", ctypes.c_int), ("cout", ctypes.c_int),
                ("transposed", ctypes.c_int)]


class SplitPackJob(ctypes.Structure):
    """ppo_split_pack_job (include/ppo_amd.h)."""
    _fields_ = [("weights", ctypes.c_void_p * 4), ("packed", ctypes.c_void_p), ("channels", ctypes.c_int),
                ("transposed", ctypes.c_int)]


class ConvPackJob(ctypes.Structure):
    """ppo_conv_pack_job (include/ppo_amd.h)."""
    _fields_ = [("weight", ctypes.c_void_p), ("packed", ctypes.c_void_p), ("cin", ctypes.c_int), ("cout", ctypes.c_int),
                ("transposed", ctypes.c_int)]


class MlpNet(ctypes.Structure):
    """ppo_mlp_net (include/ppo_amd.h)."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("w1", "b1", "w2", "b2", "wh", "bh")] + \
               [(n, ctypes.c_int) for n in ("F", "H", "NH", "act")]


class MlpGrads(ctypes.Structure):
    """ppo_mlp_grads (include/ppo_amd.h)."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("dw1", "db1", "dw2", "db2", "dwh", "dbh", "dlog_std")] + [("n_log_std", ctypes.c_int)]


class MlpLoss(ctypes.Structure):
    """ppo_mlp_loss (include/ppo_amd.h)."""
    _fields_ = [("kind", ctypes.c_int), ("grad_scale", ctypes.c_float), ("stats", ctypes.c_void_p),
                ("n_actions", ctypes.c_int), ("n_value_heads", ctypes.c_int), ("returns", ctypes.c_void_p),
                ("vf_coef", ctypes.c_float),
                ("value_col", ctypes.c_int), ("tvf_col", ctypes.c_int), ("n_tvf", ctypes.c_int), ("tvf_stride", ctypes.c_int),
                ("tvf_returns", ctypes.c_void_p), ("tvf_weights", ctypes.c_void_p), ("tvf_coef", ctypes.c_float),
                ("tvf_keep_prob", ctypes.c_float), ("seed", ctypes.c_uint64), ("offset", ctypes.c_uint64),
                ("pred_col", ctypes.c_int), ("n_pred", ctypes.c_int), ("pred_stride", ctypes.c_int),
                ("vector_targets", ctypes.c_int), ("targets", ctypes.c_void_p), ("weights", ctypes.c_void_p),
                ("old_policy", ctypes.c_void_p), ("log_std", ctypes.c_void_p), ("beta", ctypes.c_float),
                ("actions_f", ctypes.c_void_p), ("actions_i", ctypes.c_void_p), ("old_log_pac", ctypes.c_void_p),
                ("old_log_policy", ctypes.c_void_p), ("advantages", ctypes.c_void_p), ("eps_clip", ctypes.c_float),
                ("ent_coef", ctypes.c_float), ("dlog_std_rows", ctypes.c_void_p)]


MLP_LOSS_VALUE, MLP_LOSS_DISTIL, MLP_LOSS_GAUSSIAN, MLP_LOSS_PPO = 1, 2, 3, 4


class WgradJob(ctypes.Structure):
    """ppo_wgrad_job (include/ppo_amd.h)."""
    _fields_ = [("slabs", ctypes.c_void_p), ("dweight", ctypes.c_void_p), ("dbias", ctypes.c_void_p),
                ("n_slabs", ctypes.c_int), ("cin", ctypes.c_int), ("cout", ctypes.c_int), ("accumulate", ctypes.c_int)]


_lock = threading.Lock()
_lib = None


class PpoAmdError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Open libppo_amd.so and type its entry points.  Raises if it is not built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise PpoAmdError(
                f"{LIB_PATH} is missing: build it with `python -m ppo_amd.build` "
                "(hipcc, gfx950). There is no fallback path.")
        import torch  # noqa: F401  (maps PyTorch-ROCm's HIP runtime first)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().ppo_last_error()
        raise PpoAmdError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


_raw_stream = None


def current_stream() -> int:
    """Raw hipStream_t of torch's current stream on the current device.  Called once per kernel launch (tens
    of thousands of times per second), so it goes through torch's C accessor when there is one."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        fast = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        if fast is not None:
            _raw_stream = lambda: fast(torch.cuda.current_device())  # noqa: E731
        else:
            _raw_stream = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
    return _raw_stream()


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise PpoAmdError("no HIP device visible: ppo_amd runs on MI355X only, there is no CPU path")
