"""The first layer's convolution + max-pool launch (uint8 observations), training form (with argmax) and inference form,
launch time over blocks of launches, HIP events, for both kernels behind the launch (ppo_conv1_pool_form): the LDS form
(conv3x3_pool_kernel) and the form that pools out of the MFMA accumulators (conv1_pool.hip); PPO_AMD_CONV1_PRS=k forces
the latter's strip length.
COLD=1: every launch reads a different input tensor (a ring of 96 x 256 images = 690 MB: nothing of it is in L2 or the
Infinity Cache when its turn comes), as the rollout's freshly uploaded observations are.
usage (GPU box): python tools/conv1_speed.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_amd import _lib  # noqa: E402

lib = _lib.load()
st = _lib.current_stream()
for cin, hw in ((4, 84), (3, 64)):
  for n in (256, 128, 512):
    for code, form in ((1, "LDS form"), (0, "from the accumulators, strips of " + os.environ.get("PPO_AMD_CONV1_PRS", "default"))):
        lib.ppo_conv1_pool_form(code)
        w = torch.randn(16, cin, 3, 3, device="cuda") * 0.2
        b = torch.randn(16, device="cuda")
        x = torch.randint(0, 256, (n, cin, hw, hw), dtype=torch.uint8, device="cuda")
        ring = [x]
        if os.environ.get("COLD", "0") != "0":
            ring = [torch.randint(0, 256, (n, cin, hw, hw), dtype=torch.uint8, device="cuda") for _ in range(max(2, 96 * 256 // n))]
        turn = [0]
        y = torch.empty(n, 16, hw // 2, hw // 2, device="cuda")
        am = torch.empty(n, 16, hw // 2, hw // 2, dtype=torch.uint8, device="cuda")
        res = {}
        for name, amp in (("train", am.data_ptr()), ("inference", None)):
            def fn():
                turn[0] = (turn[0] + 1) % len(ring)
                return lib.ppo_conv3x3_pool_forward_f32(ring[turn[0]].data_ptr(), 2, w.data_ptr(), b.data_ptr(), y.data_ptr(), amp, n, cin, 16, hw, hw, st)
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
        flop = 2.0 * 9 * cin * 16 * hw * hw * n
        print(f"{form}: {cin}->16 {hw}x{hw} n={n}: train {res['train']:6.1f} us ({flop / res['train'] / 1e6:5.1f} TFLOP/s)   "
              f"inference {res['inference']:6.1f} us ({flop / res['inference'] / 1e6:5.1f} TFLOP/s)", flush=True)
