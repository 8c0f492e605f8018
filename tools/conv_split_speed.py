"""One stack-first convolution three ways - exact float32 MFMA (conv3x3_kernel), two-part bf16 split (three products, the
opt-in mode) and the three-part prototype (six products, float32-accurate) - launch time over blocks of launches, HIP events.
usage (GPU box): python tools/conv_split_speed.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_amd import _lib  # noqa: E402

lib = _lib.load()
st = _lib.current_stream()
n = 256
for cin, cout, hw in ((16, 32, 42), (32, 16, 42), (32, 32, 21)):
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.2
    b = torch.randn(cout, device="cuda")
    x = torch.randn(n, cin, hw, hw, device="cuda")
    y = torch.empty(n, cout, hw, hw, device="cuda")
    pks = {}
    for ns in (2, 3):
        pks[ns] = torch.zeros(int(lib.ppo_conv3x3_bf16_split_packed_bytes(cin, cout, ns)), dtype=torch.uint8, device="cuda")
        _lib.check(lib.ppo_conv3x3_pack_bf16_split(w.data_ptr(), pks[ns].data_ptr(), cin, cout, 0, ns, st), "pack")
    fns = {"f32": lambda: lib.ppo_conv3x3_forward_f32(x.data_ptr(), 0, w.data_ptr(), b.data_ptr(), None, y.data_ptr(), n, cin, cout, hw, hw, st),
           "2 parts": lambda: lib.ppo_conv3x3_bf16_split(x.data_ptr(), 0, pks[2].data_ptr(), b.data_ptr(), y.data_ptr(), n, cin, cout, hw, hw, 2, st),
           "3 parts": lambda: lib.ppo_conv3x3_bf16_split(x.data_ptr(), 0, pks[3].data_ptr(), b.data_ptr(), y.data_ptr(), n, cin, cout, hw, hw, 3, st)}
    if (cin, cout) == (32, 16):
        del fns["f32"]  # (no exact forward instance of this shape: it is a backward-data geometry)
    res = {"f32": float("nan")}
    for name, fn in fns.items():
        for _ in range(5):
            assert fn() == 0, lib.ppo_last_error()
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        res[name] = sorted(ts)[2]
    flop = 2.0 * 9 * cin * cout * hw * hw * n
    print(f"{cin}->{cout} {hw}x{hw} n={n}: exact f32 {res['f32']:6.1f} us ({flop / res['f32'] / 1e6:.0f} TFLOP/s)   two parts {res['2 parts']:6.1f} us   "
          f"three parts {res['3 parts']:6.1f} us ({res['f32'] / res['3 parts']:.2f} x the exact kernel)")
