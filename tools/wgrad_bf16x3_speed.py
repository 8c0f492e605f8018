"""Launch time of the split-bf16 weight-gradient kernel (csrc/wgrad_bf16x3.hip) against the exact float32 one on the
minibatch geometries of the Pong-shaped net (n = 256): HIP events around `reps` back-to-back launches of each.
Usage: python tools/wgrad_bf16x3_speed.py [reps]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_amd import _lib  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
lib = _lib.load()
n = 256
st = _lib.current_stream()
for cin, cout, hw, k in [(16, 16, 42, 4), (16, 32, 42, 1), (32, 32, 21, 5), (32, 32, 11, 4)]:
    xs = [torch.randn(n, cin, hw, hw, device="cuda") for _ in range(k)]
    dys = [torch.randn(n, cout, hw, hw, device="cuda") for _ in range(k)]
    ws_bytes = lib.ppo_conv3x3_wgrad_workspace_bytes(cin, cout)
    wss = [torch.empty(ws_bytes // 4, device="cuda") for _ in range(k)]
    ins = (ctypes.c_void_p * k)(*[t.data_ptr() for t in xs])
    mode = 1 if cin == cout else 0  # block convolutions read through the ReLU, a stack-first convolution reads raw
    relu = (ctypes.c_int * k)(*([mode] * k))
    dyp = (ctypes.c_void_p * k)(*[t.data_ptr() for t in dys])
    wsp = (ctypes.c_void_p * k)(*[t.data_ptr() for t in wss])
    n_slabs = ctypes.c_int(0)
    res = {}
    for name in ("f32", "bf16x3"):
        def run():
            if name == "bf16x3":
                rc = lib.ppo_conv3x3_backward_weight_slabs_batch_bf16_split(ins, relu, dyp, wsp, ws_bytes, k, n, cin, cout, hw, hw,
                                                                            int(os.environ.get("PPO_N_SPLIT", "2")),
                                                                            ctypes.addressof(n_slabs), st)
            elif k > 4:
                rc = lib.ppo_conv3x3_backward_weight_slabs_batch_mixed_f32(ins, relu, dyp, wsp, ws_bytes, k, n, cin, cout, hw, hw,
                                                                           ctypes.addressof(n_slabs), st)
            else:
                rc = lib.ppo_conv3x3_backward_weight_slabs_batch_f32(ins, mode, dyp, wsp, ws_bytes, k, n, cin, cout, hw, hw,
                                                                     ctypes.addressof(n_slabs), st)
            _lib.check(rc, name)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / reps * 1e3, n_slabs.value)
    bytes_ = k * n * (cin + cout) * hw * hw * 4
    flops = k * n * 2 * 9 * cin * cout * hw * hw
    f, b = res["f32"], res["bf16x3"]
    print(f"{cin:2d}->{cout:2d} {hw}x{hw} x{k}: f32 {f[0]:7.1f} us ({f[1]} slabs)   bf16x3 {b[0]:7.1f} us ({b[1]} slabs)   "
          f"{f[0] / b[0]:.2f} x   x+dy {bytes_ / 1e6:.0f} MB = {bytes_ / b[0] / 1e6:.2f} TB/s   {flops / b[0] / 1e6:.1f} TFLOP/s (model)")
