"""What could fusing the block weight gradients into the backward-data kernels save at most?  The batched weight-gradient
launch of a stack (four problems, one wave of workgroups) timed at the minibatch (256 images) and at batches so small that
only its per-launch fixed cost is left (prologue, K-group fold, slab writes, ramp and tail), plus the slab reduction.
usage (GPU box): python tools/wgrad_fixed_cost.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_amd import _lib  # noqa: E402

lib = _lib.load()
st = _lib.current_stream()
print("| layer | images | batched wgrad launch us | per image ns | MFMA floor us (157.3 TFLOP/s) |")
print("|---|---|---|---|---|")
for c, hw in ((16, 42), (32, 21), (32, 11)):
    nbytes = lib.ppo_conv3x3_wgrad_workspace_bytes(c, c)
    for n in (256, 64, 16, 4, 1):
        xs = [torch.randn(n, c, hw, hw, device="cuda") for _ in range(4)]
        dys = [torch.randn(n, c, hw, hw, device="cuda") for _ in range(4)]
        wss = [torch.empty(nbytes // 4 + 4, device="cuda") for _ in range(4)]
        n_slabs = ctypes.c_int(0)
        args = ((ctypes.c_void_p * 4)(*[t.data_ptr() for t in xs]), 1, (ctypes.c_void_p * 4)(*[t.data_ptr() for t in dys]),
                (ctypes.c_void_p * 4)(*[t.data_ptr() for t in wss]), nbytes, 4, n, c, c, hw, hw, ctypes.addressof(n_slabs), st)
        for _ in range(5):
            _lib.check(lib.ppo_conv3x3_backward_weight_slabs_batch_f32(*args), "wgrad batch")
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                lib.ppo_conv3x3_backward_weight_slabs_batch_f32(*args)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        t = sorted(ts)[2]
        flop = 4 * 2.0 * 9 * c * c * hw * hw * n
        print(f"| x4 {c}->{c} {hw}x{hw} | {n} | {t:.1f} | {t / n * 1e3:.0f} | {flop / 157.3e6:.1f} |")

