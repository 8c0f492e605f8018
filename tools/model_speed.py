"""Microbench: rollout forward and PPO minibatch (fwd+loss+bwd+adam) at batch 256, HIP path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd import models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
net = models.DualHeadNet("impala", (4, 84, 84), 6, hidden_units=256, head_scale=0.1, head_bias=True, device="cuda")
x = torch.randint(0, 256, (B, 4, 84, 84), dtype=torch.uint8, device="cuda")
actions = torch.randint(0, 6, (B,), device="cuda").int()
lp = torch.log_softmax(torch.randn(B, 6, device="cuda"), 1)
lpac = lp.gather(1, actions.long()[:, None])[:, 0].contiguous()
adv = torch.randn(B, device="cuda"); ret = torch.randn(B, 1, device="cuda")
def fwd(): net.forward(x)
def train():
    net.ppo_minibatch(x, actions, lpac, lp, adv, ret)
    net.adam_step()
for name, fn, reps in (("forward", fwd, 20), ("train", train, 20)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    flops = 108.4e6 * B * (1 if name == "forward" else 3)
    print(f"{name}: {dt*1e3:.3f} ms per batch of {B} -> {B/dt:.0f} samples/s, ~{flops/dt/1e12:.1f} TFLOP/s")
