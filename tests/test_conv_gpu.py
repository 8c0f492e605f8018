"""GPU: hand-written MFMA conv3x3 (forward, backward-data) against torch's fp32 conv2d.

The reference runs these contractions through torch.nn.Conv2d (rl/impala.py:61-62,96);
summation order differs between any two implementations, so the bar is the fp32
tolerance SURVEY.md §8d gives for conv results: rel 1e-4 of the tensor's max.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402

from ppo_amd import _lib  # noqa: E402

GEOMS = [(4, 16, 84), (5, 16, 84), (3, 16, 64), (16, 16, 42), (16, 32, 42), (32, 32, 21), (32, 32, 11),
         (16, 16, 32), (16, 32, 32), (32, 32, 16), (32, 32, 8)]


def _p(t):
    return None if t is None else t.data_ptr()


def _close(a, b, rel=1e-4):
    return (a.double() - b.double()).abs().max().item() <= rel * max(b.abs().max().item(), 1e-30)


def conv_fwd(x, w, b, res, mode):
    n, cin, h, wd = x.shape
    cout = w.shape[0]
    out = torch.empty((n, cout, h, wd), dtype=torch.float32, device=x.device)
    rc = _lib.load().ppo_conv3x3_forward_f32(_p(x), mode, _p(w), _p(b), _p(res), _p(out), n, cin, cout, h, wd,
                                             _lib.current_stream())
    _lib.check(rc, "ppo_conv3x3_forward_f32")
    return out


@pytest.mark.parametrize("cin,cout,hw", GEOMS)
@pytest.mark.parametrize("n", [1, 3, 37])
def test_forward_matches_torch(cin, cout, hw, n):
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(cin * 1000 + cout * 10 + hw + n)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.2).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    if cin <= 5:
        xu = torch.randint(0, 256, (n, cin, hw, hw), generator=g, dtype=torch.uint8).to(dev)
        out = conv_fwd(xu, w, b, None, _lib.PPO_IN_U8)
        ref = F.conv2d(xu.float() / 255.0, w, b, padding=1)
        assert _close(out, ref)
        xf = torch.randn(n, cin, hw, hw, generator=g).to(dev)
        assert _close(conv_fwd(xf, w, b, None, _lib.PPO_IN_NONE), F.conv2d(xf, w, b, padding=1))
    else:
        x = torch.randn(n, cin, hw, hw, generator=g).to(dev)
        res = torch.randn(n, cout, hw, hw, generator=g).to(dev)
        assert _close(conv_fwd(x, w, b, None, _lib.PPO_IN_NONE), F.conv2d(x, w, b, padding=1))
        if cin == cout:
            out = conv_fwd(x, w, b, res, _lib.PPO_IN_RELU)
            assert _close(out, F.conv2d(F.relu(x), w, b, padding=1) + res)
            assert _close(conv_fwd(x, w, None, None, _lib.PPO_IN_RELU), F.conv2d(F.relu(x), w, None, padding=1))


def test_forward_exact_on_integer_data():
    """Small-integer operands make every product and partial sum exact in fp32, so any
    operand-mapping mistake (wrong lane->element map, swapped taps) shows as a hard mismatch."""
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(7)
    for cin, cout, hw in [(16, 32, 42), (32, 32, 11), (16, 16, 42)]:
        x = torch.randint(-3, 4, (5, cin, hw, hw), generator=g).float().to(dev)
        w = torch.randint(-2, 3, (cout, cin, 3, 3), generator=g).float().to(dev)
        b = torch.randint(-5, 6, (cout,), generator=g).float().to(dev)
        assert torch.equal(conv_fwd(x, w, b, None, _lib.PPO_IN_NONE), F.conv2d(x, w, b, padding=1))


@pytest.mark.parametrize("cin,cout,hw", [g for g in GEOMS if g[0] > 5])
def test_backward_data_matches_autograd(cin, cout, hw):
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(cin + cout + hw)
    n = 9
    pre = torch.randn(n, cin, hw, hw, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.2).to(dev)
    dy = torch.randn(n, cout, hw, hw, generator=g).to(dev)
    dres = torch.randn(n, cin, hw, hw, generator=g).to(dev)
    lib = _lib.load()

    # plain: dx = conv_transpose(dy)
    y = F.conv2d(pre, w, None, padding=1)
    (ref,) = torch.autograd.grad(y, pre, dy)
    dx = torch.empty_like(ref)
    rc = lib.ppo_conv3x3_backward_data_f32(_p(dy), _p(w), None, None, _p(dx), n, cin, cout, hw, hw, _lib.current_stream())
    _lib.check(rc, "bwd")
    assert _close(dx, ref)

    # with the ReLU mask of the forward's load transform and a skip-connection gradient
    y = F.conv2d(F.relu(pre), w, None, padding=1)
    (ref,) = torch.autograd.grad(y, pre, dy)
    ref = ref + dres
    rc = lib.ppo_conv3x3_backward_data_f32(_p(dy), _p(w), _p(pre), _p(dres), _p(dx), n, cin, cout, hw, hw,
                                           _lib.current_stream())
    _lib.check(rc, "bwd")
    assert _close(dx, ref)


def test_unsupported_geometry_is_an_error():
    dev = torch.device("cuda")
    x = torch.zeros(1, 7, 10, 10, device=dev)
    w = torch.zeros(8, 7, 3, 3, device=dev)
    out = torch.zeros(1, 8, 10, 10, device=dev)
    rc = _lib.load().ppo_conv3x3_forward_f32(_p(x), 0, _p(w), None, None, _p(out), 1, 7, 8, 10, 10, _lib.current_stream())
    assert rc == -1


@pytest.mark.parametrize("cin,cout,hw,u8", [(4, 16, 84, True), (5, 16, 84, False), (3, 16, 64, True), (4, 16, 64, False),
                                            (16, 32, 42, False), (16, 32, 32, False), (32, 32, 21, False),
                                            (32, 32, 16, False)])
@pytest.mark.parametrize("n", [1, 5, 130])
@pytest.mark.parametrize("form", [-1, 0, 1])  # ppo_conv1_pool_form: by measurement / from the accumulators / LDS form
def test_fused_conv_pool_equals_conv_then_pool_bitwise(cin, cout, hw, u8, n, form):
    """ppo_conv3x3_pool_forward_f32 (stack-first conv + max-pool, the pre-pool map stays in LDS) must give
    exactly what the two separate entry points give — same MFMA accumulation order, same pooling rule
    (first maximum in row-major window order, padding excluded) — and match torch's conv2d + max_pool2d."""
    lib = _lib.load()
    dev = torch.device("cuda")
    if form != -1 and not u8:
        pytest.skip("the form switch concerns the uint8 first layer")
    before = lib.ppo_conv1_pool_form(form)
    try:
        _fused_conv_pool_case(lib, dev, cin, cout, hw, u8, n)
    finally:
        lib.ppo_conv1_pool_form(before)


def _fused_conv_pool_case(lib, dev, cin, cout, hw, u8, n):
    g = torch.Generator(device=dev).manual_seed(cin * 1000 + hw + n)
    if u8:
        x = torch.randint(0, 256, (n, cin, hw, hw), generator=g, device=dev, dtype=torch.uint8)
        mode, xf = 2, x.float() / 255.0
    else:
        x = torch.randn(n, cin, hw, hw, generator=g, device=dev)
        mode, xf = 0, x
    # quantised weights/bias make ties in the pooling windows likely (the tie rule is part of the contract)
    w = (torch.randn(cout, cin, 3, 3, generator=g, device=dev) * 4).round() / 16
    b = (torch.randn(cout, generator=g, device=dev) * 4).round() / 8
    ho = (hw + 1) // 2
    c = conv_fwd(x, w, b, None, mode)
    p_ref = torch.empty(n, cout, ho, ho, device=dev)
    i_ref = torch.empty(n, cout, ho, ho, device=dev, dtype=torch.uint8)
    _lib.check(lib.ppo_maxpool3x3s2_forward_f32(_p(c), _p(p_ref), _p(i_ref), n, cout, hw, hw, _lib.current_stream()), "pool")
    p = torch.full((n, cout, ho, ho), float("nan"), device=dev)
    i = torch.full((n, cout, ho, ho), 255, device=dev, dtype=torch.uint8)
    _lib.check(lib.ppo_conv3x3_pool_forward_f32(_p(x), mode, _p(w), _p(b), _p(p), _p(i), n, cin, cout, hw, hw,
                                                _lib.current_stream()), "ppo_conv3x3_pool_forward_f32")
    torch.cuda.synchronize()
    assert torch.equal(p, p_ref) and torch.equal(i, i_ref)
    assert _close(p, F.max_pool2d(F.conv2d(xf, w, b, padding=1), 3, 2, 1))
    # inference form: no argmax
    p2 = torch.empty_like(p)
    _lib.check(lib.ppo_conv3x3_pool_forward_f32(_p(x), mode, _p(w), _p(b), _p(p2), None, n, cin, cout, hw, hw,
                                                _lib.current_stream()), "ppo_conv3x3_pool_forward_f32")
    assert torch.equal(p2, p_ref)


@pytest.mark.parametrize("cin,cout,hw", GEOMS)
def test_packed_weights_give_the_raw_weight_results_bitwise(cin, cout, hw):
    """ppo_conv3x3_pack_weights_f32 + the *_packed_f32 entry points (forward, backward-data, fused conv+pool)
    against the raw-weight entry points on the same inputs: identical bits (only where the A operand comes from
    changes, not the arithmetic)."""
    import ctypes
    lib = _lib.load()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(cin * 31 + cout + hw)
    n = 9
    w = torch.randn(cout, cin, 3, 3, generator=g, device=dev)
    b = torch.randn(cout, generator=g, device=dev)
    pf = torch.full((lib.ppo_conv3x3_packed_floats(cin, cout, 0),), float("nan"), device=dev)
    pb = torch.full((lib.ppo_conv3x3_packed_floats(cin, cout, 1),), float("nan"), device=dev)
    jobs = (_lib.PackJob * 2)(_lib.PackJob(_p(w), _p(pf), cin, cout, 0), _lib.PackJob(_p(w), _p(pb), cin, cout, 1))
    _lib.check(lib.ppo_conv3x3_pack_weights_f32(ctypes.addressof(jobs), 2, _lib.current_stream()), "pack")
    torch.cuda.synchronize()
    assert torch.isfinite(pf).all() and torch.isfinite(pb).all()
    assert abs(float(pf.abs().sum()) - float(w.abs().sum())) < 1e-2 * float(w.abs().sum())  # a permutation + zero padding
    # forward (ReLU-on-load + residual where the geometry allows it: cin == cout)
    relu_ok = cin == cout
    x = torch.randn(n, cin, hw, hw, generator=g, device=dev)
    res = torch.randn(n, cout, hw, hw, generator=g, device=dev) if relu_ok else None
    mode = 1 if relu_ok else 0
    want = conv_fwd(x, w, b, res, mode)
    got = torch.empty_like(want)
    _lib.check(lib.ppo_conv3x3_forward_packed_f32(_p(x), mode, _p(pf), _p(b), _p(res), _p(got), n, cin, cout, hw, hw,
                                                  _lib.current_stream()), "fwd packed")
    assert torch.equal(got, want)
    # backward-data (exists for every layer that is not the observation conv)
    if cin >= 16:
        dy = torch.randn(n, cout, hw, hw, generator=g, device=dev)
        dx1, dx2 = torch.empty(n, cin, hw, hw, device=dev), torch.empty(n, cin, hw, hw, device=dev)
        _lib.check(lib.ppo_conv3x3_backward_data_f32(_p(dy), _p(w), _p(x), None, _p(dx1), n, cin, cout, hw, hw,
                                                     _lib.current_stream()), "bwd")
        _lib.check(lib.ppo_conv3x3_backward_data_packed_f32(_p(dy), _p(pb), _p(x), None, _p(dx2), n, cin, cout, hw, hw,
                                                            _lib.current_stream()), "bwd packed")
        assert torch.equal(dx1, dx2)
    # fused conv + pool (stack-first geometries)
    if (cin, cout, hw) in [(4, 16, 84), (5, 16, 84), (3, 16, 64), (16, 32, 42), (16, 32, 32), (32, 32, 21), (32, 32, 16)]:
        ho = (hw + 1) // 2
        p1, p2 = torch.empty(n, cout, ho, ho, device=dev), torch.empty(n, cout, ho, ho, device=dev)
        i1 = torch.empty(n, cout, ho, ho, device=dev, dtype=torch.uint8)
        i2 = torch.empty_like(i1)
        _lib.check(lib.ppo_conv3x3_pool_forward_f32(_p(x), 0, _p(w), _p(b), _p(p1), _p(i1), n, cin, cout, hw, hw,
                                                    _lib.current_stream()), "pool")
        _lib.check(lib.ppo_conv3x3_pool_forward_packed_f32(_p(x), 0, _p(pf), _p(b), _p(p2), _p(i2), n, cin, cout, hw, hw,
                                                           _lib.current_stream()), "pool packed")
        assert torch.equal(p1, p2) and torch.equal(i1, i2)


@pytest.mark.parametrize("cin,hw", [(4, 84), (3, 64)])  # 64x64 has no such kernel: the same cases through the LDS form
@pytest.mark.parametrize("n", [2, 100, 200, 261])  # the three strip lengths of conv1_pool.hip's launch
def test_first_layer_pooled_from_the_accumulators_special_cases(cin, hw, n):
    """conv1_pool.hip (uint8 observations, pooled out of the MFMA accumulators with DPP row shifts): the windows its
    fast path hands to the reference scan - black regions under a zero bias (a zero maximum, whose SIGN is that of the
    first zero in the window: compared as bit patterns), a channel with a NaN weight, a channel at +inf - and the
    packed-weight + minibatch-index form, all against the separate convolution and max-pool launches."""
    import ctypes
    lib = _lib.load()
    dev = torch.device("cuda")
    before = lib.ppo_conv1_pool_form(0)  # this kernel for the training form and the 64x64 maps too
    try:
        _first_layer_special_cases(lib, dev, cin, hw, n)
    finally:
        lib.ppo_conv1_pool_form(before)


def _first_layer_special_cases(lib, dev, cin, hw, n):
    import ctypes
    g = torch.Generator(device=dev).manual_seed(cin * 7 + hw + n)
    x = torch.randint(0, 256, (n, cin, hw, hw), generator=g, device=dev, dtype=torch.uint8)
    x[:, :, : hw // 3] = 0                      # a black band: rows of exact zeros (of either sign) in every channel
    x[0] = 0
    x[1 % n, :, :, hw // 2:] = 0
    w = (torch.randn(16, cin, 3, 3, generator=g, device=dev) * 4).round() / 16
    b = (torch.randn(16, generator=g, device=dev) * 4).round() / 8
    b[:6] = 0.0
    b[2] = -0.0
    w[5, 0, 1, 1] = float("nan")
    w[7, 0, 0, 2] = float("inf")
    ho = hw // 2
    c = conv_fwd(x, w, b, None, _lib.PPO_IN_U8)
    p_ref = torch.empty(n, 16, ho, ho, device=dev)
    i_ref = torch.empty(n, 16, ho, ho, device=dev, dtype=torch.uint8)
    _lib.check(lib.ppo_maxpool3x3s2_forward_f32(_p(c), _p(p_ref), _p(i_ref), n, 16, hw, hw, _lib.current_stream()), "pool")
    p = torch.full((n, 16, ho, ho), 3.0, device=dev)
    i = torch.full((n, 16, ho, ho), 255, device=dev, dtype=torch.uint8)
    _lib.check(lib.ppo_conv3x3_pool_forward_f32(_p(x), 2, _p(w), _p(b), _p(p), _p(i), n, cin, 16, hw, hw,
                                                _lib.current_stream()), "ppo_conv3x3_pool_forward_f32")
    torch.cuda.synchronize()
    finite = torch.isfinite(p_ref)
    assert torch.equal(torch.isnan(p), torch.isnan(p_ref))
    assert torch.equal(p.view(torch.int32)[~torch.isnan(p_ref)], p_ref.view(torch.int32)[~torch.isnan(p_ref)])
    assert torch.equal(i, i_ref)
    assert finite.any() and (~finite).any() and (p_ref == 0).any()
    # inference form, packed weights, images read through a permutation
    pf = torch.zeros((lib.ppo_conv3x3_packed_floats(cin, 16, 0),), device=dev)
    jobs = (_lib.PackJob * 1)(_lib.PackJob(_p(w), _p(pf), cin, 16, 0))
    _lib.check(lib.ppo_conv3x3_pack_weights_f32(ctypes.addressof(jobs), 1, _lib.current_stream()), "pack")
    perm = torch.randperm(n, generator=g, device=dev).to(torch.int32)
    for amx in (None, i):
        p2 = torch.full_like(p, 3.0)
        _lib.check(lib.ppo_conv3x3_pool_forward_packed_indexed_f32(_p(x), _p(perm), 2, _p(pf), _p(b), _p(p2), _p(amx), n, cin, 16,
                                                                   hw, hw, _lib.current_stream()), "packed indexed")
        want = p_ref[perm.long()]
        ok = ~torch.isnan(want)
        assert torch.equal(torch.isnan(p2), ~ok) and torch.equal(p2.view(torch.int32)[ok], want.view(torch.int32)[ok])
        if amx is not None:
            assert torch.equal(i, i_ref[perm.long()])


@pytest.mark.parametrize("cin,hw", [(4, 84), (3, 64)])
def test_uint8_scaling_is_the_exact_quotient_for_all_256_values(cin, hw):
    """x / 255 on load (rl/models.py:842-848) is computed as a multiply and two FMAs (csrc/conv_stage.h u8_unit); with
    a filter that copies channel 0's centre tap every output is fl(x / 255), so all 256 inputs are checked bit for
    bit against torch's division — through the plain first convolution (per-pixel staging) and through the fused
    convolution + max-pool kernel (four-pixels-per-lane staging; an image of 2x2 constant blocks pools to itself)."""
    dev = torch.device("cuda")
    lib = _lib.load()
    w = torch.zeros(16, cin, 3, 3, device=dev)
    w[0, 0, 1, 1] = 1.0
    vals = torch.arange(256, dtype=torch.uint8, device=dev)
    x = torch.zeros((2, cin, hw, hw), dtype=torch.uint8, device=dev)
    x[0, 0].view(-1)[:256] = vals
    x[1, 0].view(-1)[-256:] = vals.flip(0)
    out = conv_fwd(x, w, None, None, _lib.PPO_IN_U8)
    # the quotient is taken on the CPU: torch's GPU kernel for tensor / scalar multiplies by fl(1 / 255) instead, which
    # differs from the IEEE quotient in the last bit for 126 of the 256 inputs; the fixtures (and the reference's
    # --device=cpu path) divide
    assert torch.equal(out[:, 0].cpu(), x[:, 0].cpu().float() / 255.0)
    assert float(out[:, 1:].abs().max()) == 0.0
    # pooled: value v fills the 2x2 block b = v of channel 0 (blocks in row-major order), so pooled[b] = v / 255
    xb = torch.zeros((1, cin, hw, hw), dtype=torch.uint8, device=dev)
    blocks = torch.arange((hw // 2) ** 2, device=dev) % 256
    xb[0, 0] = blocks.to(torch.uint8).view(hw // 2, hw // 2).repeat_interleave(2, 0).repeat_interleave(2, 1)
    pooled = torch.empty((1, 16, hw // 2, hw // 2), device=dev)
    rc = lib.ppo_conv3x3_pool_forward_f32(_p(xb), _lib.PPO_IN_U8, _p(w), None, _p(pooled), None, 1, cin, 16, hw, hw,
                                          _lib.current_stream())
    _lib.check(rc, "ppo_conv3x3_pool_forward_f32")
    want = F.max_pool2d(xb[:, :1].cpu().float() / 255.0, 3, 2, 1)
    assert torch.equal(pooled[:, :1].cpu(), want)


@pytest.mark.parametrize("hw", [42, 32])
@pytest.mark.parametrize("n", [1, 3, 37, 128, 300])  # up to 256 images: half-image items on 16 waves; above: bands on 8
def test_residual_block_in_one_launch_is_the_two_convolutions_bitwise(hw, n):
    """ppo_conv3x3_block_forward_packed_f32 (rl/impala.py:66-84, the inference form for small batches: band by band, the
    intermediate map in LDS, halo rows of conv0 recomputed and rows outside the image zeroed as conv1's padding) against
    conv0 then conv1 + skip through ppo_conv3x3_forward_packed_f32: identical bits, including the image's first and last
    bands and a last band shorter than the others; and against torch within the fp32 conv tolerance."""
    import ctypes
    lib = _lib.load()
    dev = torch.device("cuda")
    assert lib.ppo_conv3x3_block_supported(16, hw, hw) == 1 and lib.ppo_conv3x3_block_supported(32, 21, 21) == 0
    g = torch.Generator(device=dev).manual_seed(hw * 7 + n)
    c = 16
    w0, w1 = (torch.randn(c, c, 3, 3, generator=g, device=dev) * 0.2 for _ in range(2))
    b0, b1 = (torch.randn(c, generator=g, device=dev) for _ in range(2))
    nf = lib.ppo_conv3x3_packed_floats(c, c, 0)
    p0, p1 = torch.empty(nf, device=dev), torch.empty(nf, device=dev)
    jobs = (_lib.PackJob * 2)(_lib.PackJob(_p(w0), _p(p0), c, c, 0), _lib.PackJob(_p(w1), _p(p1), c, c, 0))
    _lib.check(lib.ppo_conv3x3_pack_weights_f32(ctypes.addressof(jobs), 2, _lib.current_stream()), "pack")
    x = torch.randn(n, c, hw, hw, generator=g, device=dev)
    a, want = torch.empty_like(x), torch.empty_like(x)
    st = _lib.current_stream()
    _lib.check(lib.ppo_conv3x3_forward_packed_f32(_p(x), 1, _p(p0), _p(b0), None, _p(a), n, c, c, hw, hw, st), "conv0")
    _lib.check(lib.ppo_conv3x3_forward_packed_f32(_p(a), 1, _p(p1), _p(b1), _p(x), _p(want), n, c, c, hw, hw, st), "conv1")
    got = torch.full_like(x, float("nan"))
    _lib.check(lib.ppo_conv3x3_block_forward_packed_f32(_p(x), _p(p0), _p(b0), _p(p1), _p(b1), _p(got), n, c, hw, hw, st),
               "block")
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    ref = x + F.conv2d(F.relu(F.conv2d(F.relu(x), w0, b0, padding=1)), w1, b1, padding=1)
    assert _close(got, ref)
    # no kernel for other shapes: an error, not a silent wrong answer
    rc = lib.ppo_conv3x3_block_forward_packed_f32(_p(x), _p(p0), _p(b0), _p(p1), _p(b1), _p(got), n, c, hw, hw + 1, st)
    assert rc != 0
