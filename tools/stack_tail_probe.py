"""A few forwards at batch 256 (training-mode by default, `infer` as 2nd arg for inference), for rocprofv3 passes over
stack_tail_kernel (HBM traffic, kernel time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd import models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
train = not (len(sys.argv) > 2 and sys.argv[2] == "infer")
torch.manual_seed(0)
net = models.DualHeadNet("impala", (4, 84, 84), 6, hidden_units=256, head_scale=0.1, head_bias=True, device="cuda")
x = torch.randint(0, 256, (B, 4, 84, 84), dtype=torch.uint8, device="cuda")
for _ in range(6):
    net.encode(x, train=train)
torch.cuda.synchronize()
print("done")
