"""GPU: conv weight-gradient, max-pool, GEMM, policy/loss and Adam kernels against plain
PyTorch fp32 (autograd / torch.optim.Adam), the ops the reference gets from torch at
rl/impala.py:61-62,96,105, rl/models.py:84,364-366,488 and rl/rollout.py:126-141,1309-1310,1640-1753.
Tolerances: 1e-4 of the tensor's max for contractions (summation order differs), 1e-5 for
elementwise ops, exact for integer outputs (argmax, actions).
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402

from ppo_amd import _lib  # noqa: E402

DEV = "cuda"


def _p(t):
    return None if t is None else t.data_ptr()


def _close(a, b, rel=1e-4):
    return (a.double() - b.double()).abs().max().item() <= rel * max(b.abs().max().item(), 1e-30)


def _st():
    return _lib.current_stream()


# ------------------------------------------------------------------ conv weight gradient
WG = [(4, 16, 84, "u8"), (4, 16, 84, "none"), (5, 16, 84, "u8"), (3, 16, 64, "u8"), (16, 16, 42, "relu"),
      (16, 32, 42, "none"), (32, 32, 21, "relu"), (32, 32, 21, "none"), (32, 32, 11, "relu"), (16, 16, 32, "relu"),
      (16, 32, 32, "none"), (32, 32, 16, "relu"), (32, 32, 16, "none"), (32, 32, 8, "relu")]
MODE = {"none": 0, "relu": 1, "u8": 2}


@pytest.mark.parametrize("cin,cout,hw,mode", WG)
@pytest.mark.parametrize("n", [2, 41, 300])
def test_conv_weight_grad(cin, cout, hw, mode, n):
    if n == 300 and hw >= 64:
        n = 70
    lib = _lib.load()
    g = torch.Generator().manual_seed(cin * 7 + cout + hw + n)
    if mode == "u8":
        x = torch.randint(0, 256, (n, cin, hw, hw), generator=g, dtype=torch.uint8).to(DEV)
        xin = x.float() / 255.0
    else:
        x = torch.randn(n, cin, hw, hw, generator=g).to(DEV)
        xin = F.relu(x) if mode == "relu" else x
    dy = torch.randn(n, cout, hw, hw, generator=g).to(DEV)
    w = torch.zeros(cout, cin, 3, 3, device=DEV, requires_grad=True)
    b = torch.zeros(cout, device=DEV, requires_grad=True)
    y = F.conv2d(xin, w, b, padding=1)
    rw, rb = torch.autograd.grad(y, (w, b), dy)
    ws_bytes = lib.ppo_conv3x3_wgrad_workspace_bytes(cin, cout)
    ws = torch.empty(ws_bytes // 4, device=DEV)
    dw = torch.full((cout, cin, 3, 3), 7.0, device=DEV)
    db = torch.full((cout,), 7.0, device=DEV)
    rc = lib.ppo_conv3x3_backward_weight_f32(_p(x), MODE[mode], _p(dy), _p(dw), _p(db), _p(ws), ws_bytes, n, cin, cout,
                                             hw, hw, 0, _st())
    _lib.check(rc, "wgrad")
    assert _close(dw, rw) and _close(db, rb)
    # accumulate: second call adds
    rc = lib.ppo_conv3x3_backward_weight_f32(_p(x), MODE[mode], _p(dy), _p(dw), _p(db), _p(ws), ws_bytes, n, cin, cout,
                                             hw, hw, 1, _st())
    _lib.check(rc, "wgrad")
    assert _close(dw, 2 * rw) and _close(db, 2 * rb)


def test_conv_weight_grad_slabs_then_batched_reduce_is_bitwise_the_one_step_result():
    """ppo_conv3x3_backward_weight_slabs_f32 + ppo_conv3x3_wgrad_reduce_f32 (the reductions of several layers in
    one launch, incl. accumulate) against ppo_conv3x3_backward_weight_f32 on the same inputs: identical bits."""
    import ctypes
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    layers = [(16, 16, 42, 1, 37), (32, 32, 11, 1, 64), (4, 16, 84, 0, 9), (16, 32, 42, 0, 20)]
    jobs, keep, want = [], [], []
    for cin, cout, hw, mode, n in layers:
        x = torch.randn(n, cin, hw, hw, generator=g).to(DEV)
        dy = torch.randn(n, cout, hw, hw, generator=g).to(DEV)
        ws_bytes = lib.ppo_conv3x3_wgrad_workspace_bytes(cin, cout)
        ws1, ws2 = torch.empty(ws_bytes // 4, device=DEV), torch.empty(ws_bytes // 4, device=DEV)
        dw1, db1 = torch.full((cout, cin, 3, 3), 0.5, device=DEV), torch.full((cout,), -2.0, device=DEV)
        dw2, db2 = dw1.clone(), db1.clone()
        acc = 1 if cin == 32 else 0  # one layer accumulates into existing gradients
        _lib.check(lib.ppo_conv3x3_backward_weight_f32(_p(x), mode, _p(dy), _p(dw1), _p(db1), _p(ws1), ws_bytes, n, cin,
                                                       cout, hw, hw, acc, _st()), "one step")
        n_slabs = ctypes.c_int(0)
        _lib.check(lib.ppo_conv3x3_backward_weight_slabs_f32(_p(x), mode, _p(dy), _p(ws2), ws_bytes, n, cin, cout, hw, hw,
                                                             ctypes.addressof(n_slabs), _st()), "slabs")
        assert n_slabs.value > 0
        jobs.append(_lib.WgradJob(_p(ws2), _p(dw2), _p(db2), n_slabs.value, cin, cout, acc))
        keep.append((x, dy, ws2))
        want.append((dw1, db1, dw2, db2))
    table = (_lib.WgradJob * len(jobs))(*jobs)
    _lib.check(lib.ppo_conv3x3_wgrad_reduce_f32(ctypes.addressof(table), len(jobs), _st()), "reduce")
    torch.cuda.synchronize()
    for dw1, db1, dw2, db2 in want:
        assert torch.equal(dw1, dw2) and torch.equal(db1, db2)
    assert lib.ppo_conv3x3_wgrad_reduce_f32(ctypes.addressof(table), 33, _st()) == -1  # too many jobs


def test_conv_weight_grad_exact_on_integers():
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    n, cin, cout, hw = 6, 16, 32, 42
    x = torch.randint(-2, 3, (n, cin, hw, hw), generator=g).float().to(DEV)
    dy = torch.randint(-2, 3, (n, cout, hw, hw), generator=g).float().to(DEV)
    w = torch.zeros(cout, cin, 3, 3, device=DEV, requires_grad=True)
    rw, = torch.autograd.grad(F.conv2d(x, w, None, padding=1), (w,), dy)
    ws_bytes = lib.ppo_conv3x3_wgrad_workspace_bytes(cin, cout)
    ws = torch.empty(ws_bytes // 4, device=DEV)
    dw = torch.empty_like(rw)
    db = torch.empty(cout, device=DEV)
    _lib.check(lib.ppo_conv3x3_backward_weight_f32(_p(x), 0, _p(dy), _p(dw), _p(db), _p(ws), ws_bytes, n, cin, cout,
                                                   hw, hw, 0, _st()), "wgrad")
    assert torch.equal(dw, rw)
    assert torch.equal(db, dy.sum(dim=(0, 2, 3)))


# ------------------------------------------------------------------ max pool
@pytest.mark.parametrize("c,hw", [(16, 84), (32, 42), (32, 21), (16, 64), (32, 32), (32, 16), (3, 7), (2, 1)])
def test_maxpool_forward_backward(c, hw):
    lib = _lib.load()
    g = torch.Generator().manual_seed(c + hw)
    n = 5
    x = torch.randn(n, c, hw, hw, generator=g).to(DEV).requires_grad_(True)
    ref = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    ho = (hw + 1) // 2
    assert ref.shape[-1] == ho
    out = torch.empty_like(ref)
    arg = torch.empty(ref.shape, dtype=torch.uint8, device=DEV)
    _lib.check(lib.ppo_maxpool3x3s2_forward_f32(_p(x), _p(out), _p(arg), n, c, hw, hw, _st()), "pool")
    assert torch.equal(out, ref)
    dout = torch.randn(ref.shape, generator=g).to(DEV)
    (rdx,) = torch.autograd.grad(ref, x, dout)
    dx = torch.empty_like(rdx)
    _lib.check(lib.ppo_maxpool3x3s2_backward_f32(_p(dout), _p(arg), _p(dx), n, c, hw, hw, _st()), "pool bwd")
    assert _close(dx, rdx, 1e-6)


def test_maxpool_ties_route_to_first_tap_like_torch():
    lib = _lib.load()
    x = torch.zeros(1, 1, 6, 6, device=DEV, requires_grad=True)  # every window is a tie
    ref = F.max_pool2d(x, 3, 2, 1)
    dout = torch.arange(1, 10, dtype=torch.float32, device=DEV).reshape(1, 1, 3, 3)
    (rdx,) = torch.autograd.grad(ref, x, dout)
    out = torch.empty_like(ref)
    arg = torch.empty(ref.shape, dtype=torch.uint8, device=DEV)
    dx = torch.empty_like(rdx)
    _lib.check(lib.ppo_maxpool3x3s2_forward_f32(_p(x), _p(out), _p(arg), 1, 1, 6, 6, _st()), "pool")
    _lib.check(lib.ppo_maxpool3x3s2_backward_f32(_p(dout), _p(arg), _p(dx), 1, 1, 6, 6, _st()), "pool bwd")
    assert torch.equal(dx, rdx)


# ------------------------------------------------------------------ GEMM
def gemm(A, a_sm, a_sk, relu_a, B, b_sk, b_sn, relu_b, bias, mask, M, N, K, use_ws=True):
    lib = _lib.load()
    C = torch.full((M, N), float("nan"), device=DEV)
    ws_bytes = lib.ppo_gemm_workspace_bytes(M, N, K) if use_ws else 0
    ws = torch.empty(max(ws_bytes // 4, 1), device=DEV)
    rc = lib.ppo_gemm_f32(_p(A), a_sm, a_sk, relu_a, _p(B), b_sk, b_sn, relu_b, _p(bias), _p(mask), _p(C), N, M, N, K,
                          _p(ws) if use_ws else None, ws_bytes, _st())
    _lib.check(rc, "gemm")
    return C


@pytest.mark.parametrize("bsz", [1, 7, 256, 300])
def test_gemm_dense_layer_forms(bsz):
    g = torch.Generator().manual_seed(bsz)
    K, H = 3872, 256
    flat = torch.randn(bsz, K, generator=g).to(DEV)
    W = (torch.randn(H, K, generator=g) * 0.02).to(DEV)
    b = torch.randn(H, generator=g).to(DEV)
    # forward: h = relu(flat) @ W^T + b      (A k-contiguous, B k-contiguous, split-K)
    h = gemm(flat, K, 1, 1, W, 1, K, 0, b, None, bsz, H, K)
    ref = F.linear(F.relu(flat), W, b)
    assert _close(h, ref)
    assert _close(gemm(flat, K, 1, 1, W, 1, K, 0, b, None, bsz, H, K, use_ws=False), ref)
    dh = torch.randn(bsz, H, generator=g).to(DEV)
    # input gradient with ReLU gate: dflat = (dh @ W) * (flat > 0)
    dflat = gemm(dh, H, 1, 0, W, K, 1, 0, None, flat, bsz, K, H)
    assert _close(dflat, (dh @ W) * (flat > 0))
    # weight gradient: dW = dh^T @ relu(flat)
    dW = gemm(dh, 1, H, 0, flat, K, 1, 1, None, None, H, K, bsz)
    assert _close(dW, dh.t() @ F.relu(flat))


def test_gemm_heads_forms_and_colsum():
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    bsz, H, NH = 256, 256, 13
    h = torch.randn(bsz, H, generator=g).to(DEV)
    Wh = torch.randn(NH, H, generator=g).to(DEV)
    bh = torch.randn(NH, generator=g).to(DEV)
    o = gemm(h, H, 1, 1, Wh, 1, H, 0, bh, None, bsz, NH, H)
    assert _close(o, F.linear(F.relu(h), Wh, bh))
    do = torch.randn(bsz, NH, generator=g).to(DEV)
    assert _close(gemm(do, 1, NH, 0, h, H, 1, 1, None, None, NH, H, bsz), do.t() @ F.relu(h))
    assert _close(gemm(do, NH, 1, 0, Wh, H, 1, 0, None, h, bsz, H, NH), (do @ Wh) * (h > 0))
    out = torch.zeros(NH, device=DEV)
    _lib.check(lib.ppo_colsum_f32(_p(do), bsz, NH, NH, _p(out), 0, _st()), "colsum")
    assert _close(out, do.sum(0), 1e-5)
    _lib.check(lib.ppo_colsum_f32(_p(do), bsz, NH, NH, _p(out), 1, _st()), "colsum")
    assert _close(out, 2 * do.sum(0), 1e-5)


def test_gemm_exact_on_integers_asymmetric():
    g = torch.Generator().manual_seed(5)
    M, N, K = 70, 45, 37
    A = torch.randint(-3, 4, (M, K), generator=g).float().to(DEV)
    B = torch.randint(-3, 4, (K, N), generator=g).float().to(DEV)
    assert torch.equal(gemm(A, K, 1, 0, B, N, 1, 0, None, None, M, N, K), A @ B)
    At = A.t().contiguous()  # stored [K, M]
    assert torch.equal(gemm(At, 1, M, 0, B, N, 1, 0, None, None, M, N, K), A @ B)


@pytest.mark.parametrize("M,N,K", [(70, 45, 37), (130, 260, 1030), (64, 64, 256), (257, 36, 777), (33, 72, 4100)])
@pytest.mark.parametrize("layout", ["kk", "kt", "tk", "tt"])
def test_gemm_ragged_shapes_every_operand_layout(M, N, K, layout):
    """Every (k-contiguous | tile-contiguous) operand layout of the MFMA kernel on shapes that are not multiples of the
    64 x 64 x 32 tile or of 4: the scalar operand path (extent % 4 != 0), zero-padded K groups, the K-sliced form with a
    short last slice (K >= 512, N <= 512), bias + gate with and without the float4 epilogue.  Integer data: exact."""
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randint(-2, 3, (M, K), generator=g).float().to(DEV)
    B = torch.randint(-2, 3, (K, N), generator=g).float().to(DEV)
    bias = torch.randint(-4, 5, (N,), generator=g).float().to(DEV)
    mask = torch.randint(-1, 2, (M, N), generator=g).float().to(DEV)
    a_args = (A, K, 1) if layout[0] == "k" else (A.t().contiguous(), 1, M)          # [M, K] or stored [K, M]
    b_args = (B.t().contiguous(), 1, K) if layout[1] == "k" else (B, N, 1)          # stored [N, K] or [K, N]
    want = (torch.relu(A) @ B + bias) * (mask > 0)
    got = gemm(a_args[0], a_args[1], a_args[2], 1, b_args[0], b_args[1], b_args[2], 0, bias, mask, M, N, K)
    assert torch.equal(got, want)
    assert torch.equal(gemm(a_args[0], a_args[1], a_args[2], 0, b_args[0], b_args[1], b_args[2], 0, None, None, M, N, K,
                            use_ws=False), A @ B)


@pytest.mark.parametrize("bsz", [1, 5, 128, 256, 300])
def test_dense_heads_forward_is_bitwise_the_two_products(bsz):
    """ppo_dense_heads_forward_f32 (K-slice reduction of the dense product + fused heads in one launch) against the
    two ppo_gemm_f32 calls it replaces: h and the head outputs bit for bit; without a workspace it IS the two calls."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(bsz)
    K, H, NH = 3872, 256, 13
    x = torch.randn(bsz, K, generator=g).to(DEV)
    W = (torch.randn(H, K, generator=g) * 0.02).to(DEV)
    b = torch.randn(H, generator=g).to(DEV)
    Wh = (torch.randn(NH, H, generator=g) * 0.1).to(DEV)
    bh = torch.randn(NH, generator=g).to(DEV)
    h_ref = gemm(x, K, 1, 1, W, 1, K, 0, b, None, bsz, H, K)
    o_ref = gemm(h_ref, H, 1, 1, Wh, 1, H, 0, bh, None, bsz, NH, H, use_ws=False)
    ws_bytes = lib.ppo_gemm_workspace_bytes(bsz, H, K)
    ws = torch.empty(ws_bytes // 4, device=DEV)
    for use_ws in (True, False):
        h = torch.full((bsz, H), float("nan"), device=DEV)
        o = torch.full((bsz, NH), float("nan"), device=DEV)
        rc = lib.ppo_dense_heads_forward_f32(_p(x), 1, _p(W), _p(b), _p(Wh), _p(bh), 1, _p(h), _p(o), bsz, K, H, NH,
                                             _p(ws) if use_ws else None, ws_bytes if use_ws else 0, _st())
        _lib.check(rc, "dense_heads")
        if use_ws:
            assert torch.equal(h, h_ref) and torch.equal(o, o_ref)
        else:  # unsplit dense product: other summation order than the sliced reference, same as an unsplit gemm
            h2 = gemm(x, K, 1, 1, W, 1, K, 0, b, None, bsz, H, K, use_ws=False)
            assert torch.equal(h, h2) and torch.equal(o, gemm(h2, H, 1, 1, Wh, 1, H, 0, bh, None, bsz, NH, H, use_ws=False))
    assert _close(o_ref, F.relu(F.relu(x) @ W.t() + b) @ Wh.t() + bh, 1e-4)


@pytest.mark.parametrize("bsz,nA", [(128, 6), (100, 4), (1, 15), (256, 18), (37, 7), (512, 15)])
@pytest.mark.parametrize("use_ws", [True, False])
def test_dense_heads_with_action_sampling_is_bitwise_the_separate_launches(bsz, nA, use_ws):
    """ppo_dense_heads_act_forward_f32 (the rollout's action step on the finished head row, in the finalize launch) against
    ppo_dense_heads_forward_f32 followed by ppo_policy_act_f32 with the same (seed, offset): heads, raw logits, log-policy,
    sampled actions, their log-probabilities and the recorded values bit for bit - for the action counts with a fused
    form (4 / 6 / 15 / 18), one without (7: the entry point runs the separate launches itself), with and without a
    split-K workspace (without: nothing is fused), and with NULL outputs (the rollout's last row)."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(bsz * 31 + nA)
    K, H, vh = 3872, 256, 2
    NH = nA + vh + 3
    if bsz == 512:  # BASELINE configs[2]'s rollout group: 2048-wide dense layer, 2 * 15 + 1 = 31 head columns (the 32-column
        K, NH = 2048, 2 * nA + 1  # form of the finalize launch; before round 4 this shape took four launches)
    x = torch.randn(bsz, K, generator=g).to(DEV)
    W = (torch.randn(H, K, generator=g) * 0.02).to(DEV)
    b = torch.randn(H, generator=g).to(DEV)
    Wh = (torch.randn(NH, H, generator=g) * 0.1).to(DEV)
    bh = torch.randn(NH, generator=g).to(DEV)
    ws_bytes = lib.ppo_gemm_workspace_bytes(bsz, H, K) if use_ws else 0
    ws = torch.empty(max(ws_bytes // 4, 1), device=DEV)
    seed, offset = 0x1234567890ABCDEF, 987654321 * nA

    def outputs():
        return (torch.full((bsz, nA), float("nan"), device=DEV), torch.full((bsz,), -7, dtype=torch.int32, device=DEV),
                torch.full((bsz,), float("nan"), device=DEV), torch.full((bsz, nA), float("nan"), device=DEV),
                torch.full((bsz, vh), float("nan"), device=DEV))

    h1, o1 = torch.empty(bsz, H, device=DEV), torch.empty(bsz, NH, device=DEV)
    _lib.check(lib.ppo_dense_heads_forward_f32(_p(x), 1, _p(W), _p(b), _p(Wh), _p(bh), 1, _p(h1), _p(o1), bsz, K, H, NH,
                                               _p(ws) if use_ws else None, ws_bytes, _st()), "dense_heads")
    want = outputs()
    _lib.check(lib.ppo_policy_act_f32(_p(o1), bsz, NH, nA, 1.0, None, seed, offset, 0, _p(want[0]), _p(want[1]), _p(want[2]),
                                      _p(want[3]), _p(want[4]), vh, _st()), "act")
    h2, o2 = torch.full_like(h1, float("nan")), torch.full_like(o1, float("nan"))
    got = outputs()
    _lib.check(lib.ppo_dense_heads_act_forward_f32(_p(x), 1, _p(W), _p(b), _p(Wh), _p(bh), 1, _p(h2), _p(o2), bsz, K, H, NH,
                                                   _p(ws) if use_ws else None, ws_bytes, nA, 1.0, seed, offset, _p(got[0]),
                                                   _p(got[1]), _p(got[2]), _p(got[3]), _p(got[4]), vh, _st()), "dense_heads_act")
    torch.cuda.synchronize()
    assert torch.equal(h1, h2) and torch.equal(o1, o2)
    for name, a, w in zip(("log_policy", "actions", "log_pac", "raw_policy", "values"), got, want):
        assert torch.equal(a, w), name
    assert int(got[1].min()) >= 0 and int(got[1].max()) < nA and len(set(got[1].tolist())) > (1 if bsz > 8 else 0)
    # last row of a rollout: only the values are recorded
    vals = torch.full((bsz, vh), float("nan"), device=DEV)
    _lib.check(lib.ppo_dense_heads_act_forward_f32(_p(x), 1, _p(W), _p(b), _p(Wh), _p(bh), 1, _p(h2), _p(o2), bsz, K, H, NH,
                                                   _p(ws) if use_ws else None, ws_bytes, nA, 1.0, seed, offset, None, None, None,
                                                   None, _p(vals), vh, _st()), "dense_heads_act (values only)")
    assert torch.equal(vals, want[4])
    # a head row too short for the action + value columns is an error
    assert lib.ppo_dense_heads_act_forward_f32(_p(x), 1, _p(W), _p(b), _p(Wh), _p(bh), 1, _p(h2), _p(o2), bsz, K, H, nA,
                                               _p(ws) if use_ws else None, ws_bytes, nA, 1.0, seed, offset, None, None, None,
                                               None, _p(vals), vh, _st()) != 0


@pytest.mark.parametrize("bsz,H,NH,relu", [(256, 256, 13, 1), (7, 256, 13, 1), (300, 64, 4, 0), (129, 100, 16, 1), (1, 256, 1, 0)])
def test_heads_backward_one_launch(bsz, H, NH, relu):
    """ppo_heads_backward_f32: dh, dWh against the autograd formulas; dbh / db_next bit-identical to ppo_colsum_f32 of
    dheads / of the dh it wrote."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(bsz + H)
    dheads = torch.randn(bsz, NH, generator=g).to(DEV)
    h = torch.randn(bsz, H, generator=g).to(DEV)
    Wh = torch.randn(NH, H, generator=g).to(DEV)
    dh = torch.full((bsz, H), float("nan"), device=DEV)
    dWh = torch.full((NH, H), float("nan"), device=DEV)
    dbh = torch.full((NH,), float("nan"), device=DEV)
    dbn = torch.full((H,), float("nan"), device=DEV)
    rc = lib.ppo_heads_backward_f32(_p(dheads), _p(h), relu, _p(h) if relu else None, _p(Wh), _p(dh), _p(dWh), _p(dbh), _p(dbn),
                                    bsz, H, NH, _st())
    _lib.check(rc, "heads_backward")
    act = F.relu(h) if relu else h
    want_dh = (dheads @ Wh) * ((h > 0) if relu else 1.0)
    assert _close(dh, want_dh, 1e-5) and _close(dWh, dheads.t() @ act, 1e-5)
    ref = torch.empty(NH, device=DEV)
    _lib.check(lib.ppo_colsum_f32(_p(dheads), bsz, NH, NH, _p(ref), 0, _st()), "colsum")
    assert torch.equal(dbh, ref)
    ref = torch.empty(H, device=DEV)
    _lib.check(lib.ppo_colsum_f32(_p(dh), bsz, H, H, _p(ref), 0, _st()), "colsum")
    assert torch.equal(dbn, ref)
    # nullable outputs
    rc = lib.ppo_heads_backward_f32(_p(dheads), _p(h), relu, None, _p(Wh), _p(dh), _p(dWh), None, None, bsz, H, NH, _st())
    _lib.check(rc, "heads_backward")
    assert _close(dh, dheads @ Wh, 1e-5)
    assert lib.ppo_heads_backward_f32(_p(dheads), _p(h), relu, None, _p(Wh), _p(dh), _p(dWh), None, None, bsz, H, 17, _st()) != 0


def test_gemm_rejects_operands_of_2_gib_and_zero_k():
    """32-bit byte offsets inside the kernels: an operand that spans 2 GiB or more is refused (PPO_E_INVALID), nothing
    is launched.  K = 0 writes the bias (an empty sum)."""
    lib = _lib.load()
    A = torch.zeros(8, device=DEV)
    C = torch.zeros(4, 4, device=DEV)
    rc = lib.ppo_gemm_f32(_p(A), 1 << 29, 1, 0, _p(A), 1, 1 << 29, 0, None, None, _p(C), 4, 4, 4, 2, None, 0, _st())
    assert rc != 0 and b"2 GiB" in lib.ppo_last_error()
    bias = torch.arange(4.0, device=DEV)
    out = gemm(A, 0, 1, 0, A, 1, 0, 0, bias, None, 4, 4, 0)
    assert torch.equal(out, bias.expand(4, 4))


# ------------------------------------------------------------------ policy act / PPO loss
@pytest.mark.parametrize("nA", [2, 6, 15, 18])
def test_policy_act_log_softmax_gumbel_and_greedy(nA):
    lib = _lib.load()
    g = torch.Generator().manual_seed(nA)
    B, ldo = 300, 2 * nA + 1
    heads = torch.randn(B, ldo, generator=g).to(DEV)
    u = torch.rand(B, nA, generator=g).clamp_(1e-6, 1 - 1e-6).to(DEV)
    lp = torch.empty(B, nA, device=DEV)
    act = torch.empty(B, dtype=torch.int32, device=DEV)
    lpa = torch.empty(B, device=DEV)
    _lib.check(lib.ppo_policy_act_f32(_p(heads), B, ldo, nA, 1.0, _p(u), 0, 0, 0, _p(lp), _p(act), _p(lpa), None, None, 1, _st()), "act")
    ref_lp = F.log_softmax(heads[:, :nA], dim=1)
    assert _close(lp, ref_lp, 1e-5)
    scores = ref_lp - torch.log(-torch.log(u))
    top2 = scores.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4  # exclude near-ties (fp reassociation of log-sum-exp)
    assert clear.sum() > B * 0.9
    assert torch.equal(act[clear].long(), scores.argmax(1)[clear])
    assert _close(lpa, ref_lp.gather(1, act.long()[:, None])[:, 0], 1e-5)
    # greedy: exact argmax of the raw logits, first index on ties
    heads[0, :nA] = 1.0
    _lib.check(lib.ppo_policy_act_f32(_p(heads), B, ldo, nA, 1.0, None, 0, 0, 1, None, _p(act), None, None, None, 1, _st()), "act")
    assert torch.equal(act.long(), heads[:, :nA].argmax(1))
    assert act[0].item() == 0
    # internal generator: actions follow the policy distribution (chi-square-ish sanity), seeds differ
    hb = torch.zeros(20000, ldo, device=DEV)
    hb[:, :nA] = torch.linspace(0, 1.5, nA, device=DEV)
    a1 = torch.empty(20000, dtype=torch.int32, device=DEV)
    a2 = torch.empty_like(a1)
    _lib.check(lib.ppo_policy_act_f32(_p(hb), 20000, ldo, nA, 1.0, None, 123, 0, 0, None, _p(a1), None, None, None, 1, _st()), "act")
    _lib.check(lib.ppo_policy_act_f32(_p(hb), 20000, ldo, nA, 1.0, None, 124, 0, 0, None, _p(a2), None, None, None, 1, _st()), "act")
    freq = torch.bincount(a1.long(), minlength=nA).float() / 20000
    p = F.softmax(hb[0, :nA], dim=0)
    assert (freq - p).abs().max().item() < 0.02
    assert not torch.equal(a1, a2)


@pytest.mark.parametrize("nA", [4, 6, 15])
def test_ppo_loss_matches_reference_formula_autograd(nA):
    """The loss exactly as written in rl/rollout.py:1640-1660,1682,1744-1753,1596-1608, via autograd."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(nA * 3)
    B, vh = 256, 1
    ldo = 2 * nA + vh
    eps, ent, vfc, loss_scale = 0.2, 0.01, 0.5, 0.5
    heads = (torch.randn(B, ldo, generator=g) * 1.5).to(DEV).requires_grad_(True)
    actions = torch.randint(0, nA, (B,), generator=g).to(DEV)
    old_lp = F.log_softmax(torch.randn(B, nA, generator=g) * 1.5, dim=1).to(DEV)
    old_log_pac = old_lp.gather(1, actions[:, None])[:, 0] + (torch.randn(B, generator=g) * 0.3).to(DEV)
    adv = torch.randn(B, generator=g).to(DEV)
    adv[:5] = 0.0
    ret = torch.randn(B, vh, generator=g).to(DEV)

    logps = F.log_softmax(heads[:, :nA], dim=1)
    logpac = logps[range(B), actions]
    ratio = torch.exp(logpac - old_log_pac)
    clipped = torch.clamp(ratio, 1 - eps, 1 + eps)
    loss_clip = torch.min(ratio * adv, clipped * adv)
    entropy = -(logps.exp() * logps).sum(-1)
    vloss = vfc * torch.square(heads[:, nA] - ret[:, 0])
    gain = loss_clip + entropy * ent - vloss
    loss = (-gain) * loss_scale
    loss.mean().backward()

    dh = torch.full((B, ldo), float("nan"), device=DEV)
    stats = torch.empty(B, 8, device=DEV)
    act32 = actions.int()
    rc = lib.ppo_ppo_loss_f32(_p(heads), B, ldo, nA, vh, _p(act32), _p(old_log_pac), _p(old_lp), _p(adv),
                              _p(ret), eps, ent, vfc, loss_scale / B, _p(dh), _p(stats), None, _st())
    _lib.check(rc, "loss")
    # indexed form: per-sample data stays in whole-batch order, the kernel follows the permutation
    perm = torch.randperm(B, generator=g).to(DEV)
    inv = torch.argsort(perm)
    dh2 = torch.empty_like(dh)
    # row perm[b] of the big arrays holds sample b (named tensors: raw pointers must outlive the launch)
    b_act, b_lpac, b_lp, b_adv, b_ret = (t[inv].contiguous() for t in (actions.int(), old_log_pac, old_lp, adv, ret))
    perm32 = perm.int()
    rc = lib.ppo_ppo_loss_f32(_p(heads), B, ldo, nA, vh, _p(b_act), _p(b_lpac), _p(b_lp), _p(b_adv), _p(b_ret), eps, ent,
                              vfc, loss_scale / B, _p(dh2), None, _p(perm32), _st())
    _lib.check(rc, "loss indexed")
    assert torch.equal(dh, dh2)
    assert (dh - heads.grad).abs().max().item() <= 1e-5 * heads.grad.abs().max().item() + 1e-9
    assert _close(stats[:, 0], loss_clip.detach(), 1e-5)
    assert _close(stats[:, 1], entropy.detach(), 1e-5)
    assert _close(stats[:, 2], vloss.detach(), 1e-5)
    assert torch.equal(stats[:, 3], (torch.abs(ratio - 1.0) > eps).float())
    assert _close(stats[:, 4], (old_log_pac - logpac).detach(), 1e-5)
    kl = F.kl_div(old_lp, logps.detach(), log_target=True, reduction="none").sum(-1)
    assert (stats[:, 5] - kl).abs().max().item() < 1e-5
    assert _close(stats[:, 6], gain.detach(), 1e-5)


# ------------------------------------------------------------------ Adam + global-norm clip
@pytest.mark.parametrize("n,max_norm", [(1000, 20.0), (1092579, 20.0), (1092579, 0.5), (333, 0.0)])
def test_adam_with_global_norm_clip_matches_torch(n, max_norm):
    lib = _lib.load()
    g = torch.Generator().manual_seed(n)
    p_ref = torch.nn.Parameter(torch.randn(n, generator=g).to(DEV))
    opt = torch.optim.Adam([p_ref], lr=2.5e-4, eps=1e-5, betas=(0.9, 0.999))
    w = p_ref.detach().clone()
    m = torch.zeros_like(w)
    v = torch.zeros_like(w)
    ws = torch.empty(lib.ppo_adam_workspace_bytes() // 4, device=DEV)
    norm = torch.zeros(1, device=DEV)
    for step in range(1, 6):
        grad = (torch.randn(n, generator=g) * (0.05 * step)).to(DEV)
        p_ref.grad = grad.clone()
        ref_norm = grad.norm(2)
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_([p_ref], max_norm)
        opt.step()
        rc = lib.ppo_adam_step_f32(_p(w), _p(grad), _p(m), _p(v), n, step, 2.5e-4, 0.9, 0.999, 1e-5, max_norm, 1.0,
                                   _p(ws), _p(norm), _st())
        _lib.check(rc, "adam")
        assert abs(norm.item() - ref_norm.item()) <= 1e-5 * ref_norm.item()
        assert (w - p_ref.detach()).abs().max().item() <= 2e-6 * max(1.0, p_ref.detach().abs().max().item())
    st = opt.state[p_ref]
    assert _close(m, st["exp_avg"], 1e-5) and _close(v, st["exp_avg_sq"], 1e-5)


def test_adam_grad_div_equals_prescaled_gradients():
    lib = _lib.load()
    g = torch.Generator().manual_seed(9)
    n = 5000
    w0 = torch.randn(n, generator=g).to(DEV)
    grad = torch.randn(n, generator=g).to(DEV)
    ws = torch.empty(lib.ppo_adam_workspace_bytes() // 4, device=DEV)
    outs = []
    for gsc, div in ((1.0, 1.0), (8.0, 8.0)):
        w, m, v = w0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        gg = grad * gsc
        _lib.check(lib.ppo_adam_step_f32(_p(w), _p(gg), _p(m), _p(v), n, 1, 2.5e-4, 0.9, 0.999, 1e-5, 0.5, div, _p(ws),
                                         None, _st()), "adam")
        outs.append(w)
    assert (outs[0] - outs[1]).abs().max().item() < 1e-7


@pytest.mark.parametrize("bsz,nA", [(256, 6), (100, 4), (64, 15), (37, 7)])
def test_dense_heads_with_the_ppo_loss_is_bitwise_the_separate_launches(bsz, nA):
    """ppo_dense_heads_loss_forward_f32 (the discrete PPO loss on the finished head row, in the finalize launch of a training
    forward) against ppo_dense_heads_forward_f32 followed by ppo_ppo_loss_f32: heads, d loss / d heads and the statistics
    rows bit for bit, through a minibatch index, for action counts with a fused form and one without (7)."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(bsz * 7 + nA)
    K, H, vh = 3872, 256, 1
    NH = 2 * nA + vh
    Btot = bsz + 50
    x = torch.randn(bsz, K, generator=g).to(DEV)
    W = (torch.randn(H, K, generator=g) * 0.02).to(DEV)
    b = torch.randn(H, generator=g).to(DEV)
    Wh = (torch.randn(NH, H, generator=g) * 0.1).to(DEV)
    bh = torch.randn(NH, generator=g).to(DEV)
    idx = torch.randperm(Btot, generator=g)[:bsz].int().to(DEV)
    actions = torch.randint(0, nA, (Btot,), generator=g).int().to(DEV)
    lp = torch.log_softmax(torch.randn(Btot, nA, generator=g), dim=1).to(DEV)
    pac = lp.gather(1, actions.long()[:, None])[:, 0].contiguous()
    adv, ret = torch.randn(Btot, generator=g).to(DEV), torch.randn(Btot, vh, generator=g).to(DEV)
    ws_bytes = lib.ppo_gemm_workspace_bytes(bsz, H, K)
    ws = torch.empty(max(ws_bytes // 4, 1), device=DEV)
    h1, o1 = torch.empty(bsz, H, device=DEV), torch.empty(bsz, NH, device=DEV)
    d1, s1 = torch.full((bsz, NH), float("nan"), device=DEV), torch.full((bsz, 8), float("nan"), device=DEV)
    _lib.check(lib.ppo_dense_heads_forward_f32(_p(x), 1, _p(W), _p(b), _p(Wh), _p(bh), 1, _p(h1), _p(o1), bsz, K, H, NH, _p(ws),
                                               ws_bytes, _st()), "dense_heads")
    loss = (nA, vh, _p(actions), _p(pac), _p(lp), _p(adv), _p(ret), 0.2, 0.01, 0.5, 1.0 / bsz)
    _lib.check(lib.ppo_ppo_loss_f32(_p(o1), bsz, NH, *loss, _p(d1), _p(s1), _p(idx), _st()), "ppo_loss")
    h2, o2 = torch.full_like(h1, float("nan")), torch.full_like(o1, float("nan"))
    d2, s2 = torch.full_like(d1, float("nan")), torch.full_like(s1, float("nan"))
    _lib.check(lib.ppo_dense_heads_loss_forward_f32(_p(x), 1, _p(W), _p(b), _p(Wh), _p(bh), 1, _p(h2), _p(o2), bsz, K, H, NH, _p(ws),
                                                    ws_bytes, *loss, _p(d2), _p(s2), _p(idx), _st()), "dense_heads_loss")
    torch.cuda.synchronize()
    assert torch.equal(h1, h2) and torch.equal(o1, o2) and torch.equal(d1, d2) and torch.equal(s1, s2)
    assert torch.isfinite(d2).all() and float(d2.abs().max()) > 0
