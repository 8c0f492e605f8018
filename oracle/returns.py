"""ctypes front-end of oracle/returns_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT.

Same call signatures as the reference's rl/returns.py (gae :7, td_lambda :58,
calculate_bootstrapped_returns :32) so parity tests read like the reference's
own.  NumPy in, NumPy out, host only.
"""
import ctypes

import numpy as np

from . import lib

TERM_NONE, TERM_BOOL, TERM_F32 = 0, 1, 2

_c_f32p = ctypes.POINTER(ctypes.c_float)


def _f32(x, shape=None):
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    if shape is not None:
        x = np.ascontiguousarray(np.broadcast_to(x, shape))
    return x


def _term(term):
    """bool -> uint8 (f64 carry in the reference); anything else -> f32."""
    if term is None:
        return None, TERM_NONE
    term = np.asarray(term)
    if term.dtype == np.bool_:
        return np.ascontiguousarray(term.astype(np.uint8)), TERM_BOOL
    return np.ascontiguousarray(term.astype(np.float32)), TERM_F32


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def gae(batch_rewards, batch_value, final_value_estimate, batch_terminal, gamma, lamb=0.95):
    r = _f32(batch_rewards)
    N, A = r.shape
    v = _f32(batch_value, (N, A))
    vf = _f32(final_value_estimate, (A,))
    t, kind = _term(batch_terminal)
    out = np.empty((N, A), np.float32)
    rc = lib().oracle_gae(_ptr(r), _ptr(v), _ptr(vf), _ptr(t), kind, N, A, A,
                          ctypes.c_double(gamma), ctypes.c_double(lamb), _ptr(out))
    assert rc == 0
    return out


def td_lambda(batch_rewards, batch_value, final_value_estimate, batch_terminal, gamma, lamb=0.95):
    r = _f32(batch_rewards)
    N, A = r.shape
    v = _f32(batch_value, (N, A))
    vf = _f32(final_value_estimate, (A,))
    t, kind = _term(batch_terminal)
    out = np.empty((N, A), np.float32)
    rc = lib().oracle_td_lambda(_ptr(r), _ptr(v), _ptr(vf), _ptr(t), kind, N, A, A,
                                ctypes.c_double(gamma), ctypes.c_double(lamb), _ptr(out))
    assert rc == 0
    return out


def gae_and_returns(batch_rewards, batch_value, final_value_estimate, batch_terminal,
                    gamma, lam_adv, lam_ret):
    """The pair Runner.calculate_returns computes (rl/rollout.py:1207-1223)."""
    r = _f32(batch_rewards)
    N, A = r.shape
    v = _f32(batch_value, (N, A))
    vf = _f32(final_value_estimate, (A,))
    t, kind = _term(batch_terminal)
    adv = np.empty((N, A), np.float32)
    ret = np.empty((N, A), np.float32)
    rc = lib().oracle_gae_and_returns(
        _ptr(r), _ptr(v), _ptr(vf), _ptr(t), kind, N, A, A, ctypes.c_double(gamma),
        ctypes.c_double(lam_adv), ctypes.c_double(lam_ret), _ptr(adv), _ptr(ret))
    assert rc == 0
    return adv, ret


def calculate_bootstrapped_returns(rewards, dones, final_value_estimate, gamma):
    r = _f32(rewards)
    N, A = r.shape
    vf = _f32(final_value_estimate, (A,))
    d, kind = _term(dones)
    assert kind != TERM_NONE, "the reference does not accept dones=None here"
    garr = None
    g = 0.0
    if type(gamma) is float:  # the reference's own test (rl/returns.py:49)
        g = gamma
    else:
        garr = _f32(gamma, (N, A))
    out = np.empty((N, A), np.float32)
    rc = lib().oracle_bootstrapped_returns(
        _ptr(r), _ptr(d), kind, _ptr(vf), _ptr(garr), ctypes.c_double(g), N, A, A, _ptr(out))
    assert rc == 0
    return out
