"""Debug: compare every intermediate gradient of the HIP backward with torch autograd (fp64) on GPU."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from ppo_amd import models

g = np.load("tests/golden/model_golden.npz"); meta = json.load(open("tests/golden/model_golden.json"))
torch.manual_seed(1)
net = models.DualHeadNet("impala", (4, 84, 84), 6, hidden_units=256, head_scale=0.1, head_bias=True, device="cuda")
dt = torch.float64
sd = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in net.params.items()}
x = torch.from_numpy(g["mb0_prev_state"]).cuda()
keep = {}
def fwd(x):
    for si in range(3):
        p = f"encoder.stacks.{si}."
        c = F.conv2d(x, sd[p + "firstconv.weight"], sd[p + "firstconv.bias"], padding=1); c.retain_grad(); keep[f"c{si}"] = c
        x = F.max_pool2d(c, 3, 2, 1); x.retain_grad(); keep[f"p{si}"] = x
        for bi in range(2):
            b = p + f"blocks.{bi}."
            a = F.conv2d(F.relu(x), sd[b + "conv0.weight"], sd[b + "conv0.bias"], padding=1); a.retain_grad(); keep[f"a{si}_{bi}"] = a
            r = F.conv2d(F.relu(a), sd[b + "conv1.weight"], sd[b + "conv1.bias"], padding=1)
            x = x + r; x.retain_grad(); keep[f"q{si}_{bi}"] = x
    flat = x.reshape(x.shape[0], -1)
    h = F.linear(F.relu(flat), sd["encoder.dense.weight"], sd["encoder.dense.bias"]); h.retain_grad(); keep["h"] = h
    f = F.relu(h)
    o = torch.cat([F.linear(f, sd["policy_head.weight"], sd["policy_head.bias"]), F.linear(f, sd["value_head.weight"], sd["value_head.bias"]),
                   F.linear(f, sd["advantage_head.weight"], sd["advantage_head.bias"])], 1)
    return o
o = fwd(x.to(dt) / 255.0)
acts = net.encode(x, train=True)
oh = net.heads(acts, "t")
print("fwd heads err", (oh.double() - o).abs().max().item() / o.abs().max().item())
for k in ["c0", "p0", "a0_0", "q0_1", "c1", "q1_1", "c2", "q2_1"]:
    pass
torch.manual_seed(0)
dheads = torch.randn(16, 13, device="cuda") * 0.01
dheads[:, 7:] = 0
o.backward(dheads.to(dt))
net.backward(acts, dheads)
torch.cuda.synchronize()
def rel(a, b): return ((a.double() - b).abs().max() / b.abs().max()).item()
B = 16
def buf(name, shape): return net._bufs[(name, tuple(shape), torch.float32)]
print("dh", rel(buf("dh", (B, 256)), keep["h"].grad))
# argmax agreement
for si, (c, hw) in enumerate([(16, 84), (32, 42), (32, 21)]):
    ho = (hw + 1) // 2
    idx = buf(f"tidx{si}", (B, c, ho, ho)) if (f"tidx{si}", (B, c, ho, ho), torch.uint8) not in net._bufs else net._bufs[(f"tidx{si}", (B, c, ho, ho), torch.uint8)]
    _, ref_idx = F.max_pool2d(keep[f"c{si}"].detach(), 3, 2, 1, return_indices=True)
    oy = torch.arange(ho, device="cuda")[:, None]; ox = torch.arange(ho, device="cuda")[None, :]
    ky = idx.long() // 3; kx = idx.long() % 3
    mine = (2 * oy - 1 + ky) * hw + (2 * ox - 1 + kx)
    print(f"stack {si}: argmax mismatches {(mine != ref_idx).sum().item()} of {mine.numel()}")
    print(f"  dc{si}", rel(buf(f"g{si}_dc", (B, c, hw, hw)), keep[f"c{si}"].grad))
for name in ["encoder.stacks.2.blocks.1.conv1.weight", "encoder.stacks.2.blocks.0.conv0.weight", "encoder.stacks.2.firstconv.weight",
             "encoder.stacks.1.blocks.1.conv1.weight", "encoder.stacks.1.firstconv.weight", "encoder.stacks.0.blocks.1.conv1.weight",
             "encoder.stacks.0.blocks.0.conv0.weight", "encoder.stacks.0.firstconv.weight", "encoder.stacks.0.firstconv.bias", "encoder.dense.weight"]:
    print(name, rel(net.grads[name], sd[name].grad))
print("---- stack 1 detail")
print("g wrt p1 (buffer g1_a)", rel(buf("g1_a", (B, 32, 21, 21)), keep["p1"].grad))
print("g wrt q1_0 (buffer g1_b)", rel(buf("g1_b", (B, 32, 21, 21)), keep["q1_0"].grad))
print("g wrt p2 (g2_a)", rel(buf("g2_a", (B, 32, 11, 11)), keep["p2"].grad), " q2_0 (g2_b)", rel(buf("g2_b", (B, 32, 11, 11)), keep["q2_0"].grad))
from ppo_amd import _lib
lib = _lib.load()
gp1 = keep["p1"].grad.float().contiguous()
idx1 = net._bufs[("tidx1", (B, 32, 21, 21), torch.uint8)]
out = torch.empty(B, 32, 42, 42, device="cuda")
lib.ppo_maxpool3x3s2_backward_f32(gp1.data_ptr(), idx1.data_ptr(), out.data_ptr(), B, 32, 42, 42, None)
torch.cuda.synchronize()
print("maxpool_bwd on exact g_p1:", rel(out, keep["c1"].grad))
print("---- standalone bwd_data checks at 21x21 with exact inputs")
gq = keep["q1_1"].grad.float().contiguous()        # grad wrt block-1 output of stack 1
a11 = net._bufs[("ta1_1", (B, 32, 21, 21), torch.float32)]
w1 = net.params["encoder.stacks.1.blocks.1.conv1.weight"]
da = torch.empty(B, 32, 21, 21, device="cuda")
lib.ppo_conv3x3_backward_data_f32(gq.data_ptr(), w1.data_ptr(), a11.data_ptr(), None, da.data_ptr(), B, 32, 32, 21, 21, None)
torch.cuda.synchronize()
ref = keep["a1_1"].grad
err = (da.double() - ref).abs()
print("da standalone:", (err.max() / ref.abs().max()).item())
pos = (err > 1e-3 * ref.abs().max()).nonzero()
print("bad positions:", pos.shape[0], "of", err.numel())
if pos.shape[0]:
    print("rows hist", torch.bincount(pos[:, 2], minlength=21).tolist())
    print("cols hist", torch.bincount(pos[:, 3], minlength=21).tolist())
    print("chan hist", torch.bincount(pos[:, 1], minlength=32).tolist())
    print("img hist", torch.bincount(pos[:, 0], minlength=B).tolist())
# same through torch autograd in fp32 for the op alone
pre = a11.clone().requires_grad_(True)
y = F.conv2d(F.relu(pre), w1, None, padding=1)
(r32,) = torch.autograd.grad(y, pre, gq)
print("torch fp32 op-alone vs fp64 chain:", rel(r32, ref))
print("mine vs torch fp32 op-alone:", ((da - r32).abs().max() / r32.abs().max()).item())
print("buffer g1_da (holds block-0 da) vs a1_0.grad", rel(buf("g1_da", (B, 32, 21, 21)), keep["a1_0"].grad))
