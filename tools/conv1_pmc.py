"""A few launches of the first layer's convolution + max-pool (n = 256: training form, then inference form) for a
rocprofv3 --pmc pass: rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU ... --output-format csv -d <out> -- python3 tools/conv1_pmc.py (PPO_AMD_CONV1_LDS=0|1 picks the kernel)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_amd import _lib  # noqa: E402

lib = _lib.load()
st = _lib.current_stream()
n, cin, hw = 256, 4, 84
w = torch.randn(16, cin, 3, 3, device="cuda") * 0.2
b = torch.randn(16, device="cuda")
x = torch.randint(0, 256, (n, cin, hw, hw), dtype=torch.uint8, device="cuda")
y = torch.empty(n, 16, hw // 2, hw // 2, device="cuda")
am = torch.empty(n, 16, hw // 2, hw // 2, dtype=torch.uint8, device="cuda")
for amp in (am.data_ptr(), None):
    for _ in range(3):
        assert lib.ppo_conv3x3_pool_forward_f32(x.data_ptr(), 2, w.data_ptr(), b.data_ptr(), y.data_ptr(), amp, n, cin, 16, hw, hw, st) == 0
torch.cuda.synchronize()
