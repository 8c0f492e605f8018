"""A few training-mode forwards at batch 256, for rocprofv3 --pmc passes over stack_tail_kernel (HBM traffic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd import models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
net = models.DualHeadNet("impala", (4, 84, 84), 6, hidden_units=256, head_scale=0.1, head_bias=True, device="cuda")
x = torch.randint(0, 256, (B, 4, 84, 84), dtype=torch.uint8, device="cuda")
for _ in range(6):
    net.encode(x, train=True)
torch.cuda.synchronize()
print("done")
