"""GAE / lambda-returns on the GPU — host mirror of the reference's rl/returns.py.

Same names, argument order and meaning as the reference:
``gae`` (rl/returns.py:7), ``td_lambda`` (:58), ``calculate_bootstrapped_returns``
(:32).  ``gae_and_returns`` is the fused pair ``Runner.calculate_returns``
needs (rl/rollout.py:1207-1223): one pass over the rollout writes both the
advantages (lambda_policy) and the value targets (lambda_value).

Inputs may be NumPy arrays (uploaded, result downloaded as NumPy — the
reference's calling convention) or torch tensors on the GPU (result stays on
the GPU, nothing is synchronised).  Inputs are never modified.  All compute
happens in libppo_amd.so; there is no CPU path.
"""
import numpy as np
import torch

from . import _lib

__all__ = ["gae", "td_lambda", "gae_and_returns", "calculate_bootstrapped_returns"]


def _device():
    _lib.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def _to_dev_f32(x, shape=None):
    dev = _device()
    if isinstance(x, torch.Tensor):
        t = x.to(device=dev, dtype=torch.float32)
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32))).to(dev)
    if shape is not None and tuple(t.shape) != tuple(shape):
        t = t.expand(shape)
    return t


def _rows(t):
    """[N, A] tensor with unit column stride -> (tensor, ld)."""
    if t.dim() != 2:
        raise ValueError(f"expected a [N, A] array, got shape {tuple(t.shape)}")
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def _terminals(term, shape):
    """NumPy/torch bool -> U8 (float64 recurrence), other dtypes -> F32, None -> NONE."""
    if term is None:
        return None, _lib.PPO_TERM_NONE
    dev = _device()
    if isinstance(term, torch.Tensor):
        if term.dtype == torch.bool:
            t = term.to(dev).contiguous().view(torch.uint8)
            kind = _lib.PPO_TERM_U8
        else:
            t = term.to(device=dev, dtype=torch.float32).contiguous()
            kind = _lib.PPO_TERM_F32
    else:
        term = np.asarray(term)
        if term.dtype == np.bool_:
            t = torch.from_numpy(np.ascontiguousarray(term).view(np.uint8)).to(dev)
            kind = _lib.PPO_TERM_U8
        else:
            t = torch.from_numpy(np.ascontiguousarray(term.astype(np.float32))).to(dev)
            kind = _lib.PPO_TERM_F32
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"terminals shape {tuple(t.shape)} != rewards shape {tuple(shape)}")
    return t, kind


def _ptr(t):
    return None if t is None else t.data_ptr()


def _scan(batch_rewards, batch_value, final_value_estimate, batch_terminal, gamma, lam_adv, lam_ret,
          want_adv, want_ret, regime=_lib.PPO_SCAN_AUTO):
    as_numpy = not isinstance(batch_rewards, torch.Tensor)
    r = _rows(_to_dev_f32(batch_rewards))
    N, A = r.shape
    v = _to_dev_f32(batch_value, (N, A))
    v = _rows(v)
    vf = _to_dev_f32(final_value_estimate)
    vf = (vf.expand(A) if vf.dim() == 0 else vf.reshape(-1)).contiguous()
    if vf.numel() != A:
        raise ValueError(f"final_value_estimate has {vf.numel()} entries, expected {A}")
    term, kind = _terminals(batch_terminal, (N, A))
    # one leading dimension for every operand: make the rare strided one contiguous
    ld = A
    if r.stride(0) != A and N > 1:
        r = r.contiguous()
    if v.stride(0) != A and N > 1:
        v = v.contiguous()
    adv = torch.empty((N, A), dtype=torch.float32, device=r.device) if want_adv else None
    ret = torch.empty((N, A), dtype=torch.float32, device=r.device) if want_ret else None
    lib = _lib.load()
    if N > 0 and A > 0:
        rc = lib.ppo_gae_scan_f32(_ptr(r), _ptr(v), _ptr(vf), _ptr(term), kind, _ptr(adv), _ptr(ret),
                                  N, A, ld, float(gamma), float(lam_adv), float(lam_ret), regime,
                                  _lib.current_stream())
        _lib.check(rc, "ppo_gae_scan_f32")
    if as_numpy:
        adv = None if adv is None else adv.cpu().numpy()
        ret = None if ret is None else ret.cpu().numpy()
    return adv, ret


def gae(batch_rewards, batch_value, final_value_estimate, batch_terminal, gamma: float, lamb=0.95,
        regime=_lib.PPO_SCAN_AUTO):
    """Generalised advantage estimates, [N, A] float32 (reference: rl/returns.py:7-29)."""
    return _scan(batch_rewards, batch_value, final_value_estimate, batch_terminal, gamma, lamb, lamb,
                 True, False, regime)[0]


def td_lambda(batch_rewards, batch_value, final_value_estimate, batch_terminal, gamma: float, lamb=0.95,
              regime=_lib.PPO_SCAN_AUTO):
    """TD(lambda) returns = gae + value (reference: rl/returns.py:58-67)."""
    return _scan(batch_rewards, batch_value, final_value_estimate, batch_terminal, gamma, lamb, lamb,
                 False, True, regime)[1]


def gae_and_returns(batch_rewards, batch_value, final_value_estimate, batch_terminal, gamma: float,
                    lam_adv=0.95, lam_ret=0.95, regime=_lib.PPO_SCAN_AUTO):
    """(gae(lam_adv), td_lambda(lam_ret)) in one pass over the rollout."""
    return _scan(batch_rewards, batch_value, final_value_estimate, batch_terminal, gamma, lam_adv,
                 lam_ret, True, True, regime)


def calculate_bootstrapped_returns(rewards, dones, final_value_estimate, gamma):
    """Discounted returns bootstrapped from a final value (reference: rl/returns.py:32-55).

    ``gamma`` is a python float or an [N, A] array, as in the reference.
    """
    as_numpy = not isinstance(rewards, torch.Tensor)
    r = _rows(_to_dev_f32(rewards))
    if r.stride(0) != r.shape[1] and r.shape[0] > 1:
        r = r.contiguous()
    N, A = r.shape
    vf = _to_dev_f32(final_value_estimate)
    vf = (vf.expand(A) if vf.dim() == 0 else vf.reshape(-1)).contiguous()
    d, kind = _terminals(dones, (N, A))
    if kind == _lib.PPO_TERM_NONE:
        raise TypeError("dones must be an array (the reference evaluates 1.0 - dones[i])")
    garr = None
    g = 0.0
    if type(gamma) is float:  # same test as rl/returns.py:49
        g = gamma
    else:
        garr = _to_dev_f32(gamma, (N, A)).contiguous()
    out = torch.empty((N, A), dtype=torch.float32, device=r.device)
    if N > 0 and A > 0:
        rc = _lib.load().ppo_bootstrapped_returns_f32(_ptr(r), _ptr(d), kind, _ptr(vf), _ptr(garr), g,
                                                      _ptr(out), N, A, A, _lib.current_stream())
        _lib.check(rc, "ppo_bootstrapped_returns_f32")
    return out.cpu().numpy() if as_numpy else out
