"""Time ppo_gemm_f32 at the shapes the IMPALA model uses (batch 256 / 128): dense forward, dense dW, dense dX,
heads forward / dW / dX.  Usage: python tools/gemm_speed.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_amd import _lib  # noqa: E402

lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
F, H, NH = 3872, 256, 13
dev = torch.device("cuda")
x = torch.randn(B, F, device=dev)
w = torch.randn(H, F, device=dev)
h = torch.randn(B, H, device=dev)
dh = torch.randn(B, H, device=dev)
wh = torch.randn(NH, H, device=dev)
dheads = torch.randn(B, NH, device=dev)
ws_bytes = lib.ppo_gemm_workspace_bytes(B, F, F)
ws = torch.empty(ws_bytes // 4 + 16, device=dev)


def p(t):
    return None if t is None else t.data_ptr()


def gemm(A, a_sm, a_sk, relu_a, Bm, b_sk, b_sn, relu_b, bias, mask, C, ldc, M, N, K, use_ws=True):
    rc = lib.ppo_gemm_f32(p(A), a_sm, a_sk, relu_a, p(Bm), b_sk, b_sn, relu_b, p(bias), p(mask), p(C), ldc, M, N, K,
                          p(ws) if use_ws else None, ws_bytes if use_ws else 0, _lib.current_stream())
    _lib.check(rc, "ppo_gemm_f32")


out_h = torch.empty(B, H, device=dev)
gw = torch.empty(H, F, device=dev)
gx = torch.empty(B, F, device=dev)
oh = torch.empty(B, NH, device=dev)
gwh = torch.empty(NH, H, device=dev)
cases = {
    f"dense fwd   [{B}x{F}]x[{F}x{H}]": (lambda: gemm(x, F, 1, 1, w, 1, F, 0, None, None, out_h, H, B, H, F), 2 * B * H * F),
    f"dense dW    [{H}x{B}]x[{B}x{F}]": (lambda: gemm(dh, 1, H, 0, x, F, 1, 1, None, None, gw, F, H, F, B, False), 2 * B * H * F),
    f"dense dX    [{B}x{H}]x[{H}x{F}]": (lambda: gemm(dh, H, 1, 0, w, F, 1, 0, None, x, gx, F, B, F, H, False), 2 * B * H * F),
    f"heads fwd   [{B}x{H}]x[{H}x{NH}]": (lambda: gemm(h, H, 1, 1, wh, 1, H, 0, None, None, oh, NH, B, NH, H, False), 2 * B * H * NH),
    f"heads dW    [{NH}x{B}]x[{B}x{H}]": (lambda: gemm(dheads, 1, NH, 0, h, H, 1, 1, None, None, gwh, H, NH, H, B, False), 2 * B * H * NH),
    f"heads dX    [{B}x{NH}]x[{NH}x{H}]": (lambda: gemm(dheads, NH, 1, 0, wh, H, 1, 0, None, h, out_h, H, B, H, NH, False), 2 * B * H * NH),
}
for name, (fn, flops) in cases.items():
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 200
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name:40s} {us:8.1f} us  {flops / us / 1e6:7.2f} TFLOP/s")
