"""Interleaved A/B of the residual-block launch of a 32-channel stack: exact float32 (stack_tail_kernel) against the
opt-in split-bf16 form (csrc/stack_bf16x3.hip), inference geometry (q1 only), HIP events on the launch stream.
usage (GPU box): python tools/bf16x3_ab.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_amd import _lib, models  # noqa: E402

lib = _lib.load()
torch.manual_seed(0)
net = models.DualHeadNet("impala", (4, 84, 84), 6, hidden_units=256, head_scale=0.1, head_bias=True, device="cuda")
net._refresh_packed()
st = _lib.current_stream()
print("| map | images | exact f32 us | bf16 x 3 us | ratio | f32 TFLOP/s (frac of 157.3) | bf16x3 effective TFLOP/s |")
print("|---|---|---|---|---|---|---|")
for hw, stack in ((21, 1), (11, 2)):
    names = [f"encoder.stacks.{stack}.blocks.{bi}.conv{ci}" for bi in range(2) for ci in range(2)]
    ws = [net.params[n + ".weight"] for n in names]
    bs = [net.params[n + ".bias"] for n in names]
    packed = torch.empty(int(lib.ppo_impala_stack_tail_bf16x3_packed_bytes()), dtype=torch.uint8, device="cuda")
    wp = (ctypes.c_void_p * 4)(*[w.data_ptr() for w in ws])
    bp = (ctypes.c_void_p * 4)(*[b.data_ptr() for b in bs])
    _lib.check(lib.ppo_impala_stack_tail_pack_bf16x3(wp, packed.data_ptr(), 32, 0, st), "pack")
    tail = net._stack_tail_ptrs(stack, 32, hw, hw)
    for B in (128, 256, 1024):
        p = torch.randn(B, 32, hw, hw, device="cuda")
        q = torch.empty_like(p)

        def f32():
            lib.ppo_impala_stack_tail_forward_f32(p.data_ptr(), tail[0], tail[1], None, None, None, q.data_ptr(), B, 32, hw, hw, st)

        def b16():
            lib.ppo_impala_stack_tail_forward_bf16x3(p.data_ptr(), packed.data_ptr(), bp, None, None, None, q.data_ptr(), B, 32, hw, hw, st)
        times = {"f32": [], "b16": []}
        for fn in (f32, b16):
            for _ in range(5):
                fn()
        torch.cuda.synchronize()
        for rep in range(7):  # alternate blocks of 20 launches
            for name, fn in (("f32", f32), ("b16", b16)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[name].append(e0.elapsed_time(e1) / 20 * 1e3)
        a, b = sorted(times["f32"])[3], sorted(times["b16"])[3]
        flop = 4 * 2.0 * 9 * 32 * 32 * hw * hw * B
        print(f"| {hw}x{hw} | {B} | {a:.1f} | {b:.1f} | {a / b:.2f} | {flop / a / 1e6:.1f} ({flop / a / 1e6 / 157.3:.2f}) | {flop / b / 1e6:.1f} |")
