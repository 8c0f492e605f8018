"""Launch times of the split-bf16 residual-block launches in their TRAINING forms (forward keeping a0 / q0 / a1, gated
backward-data chain) and the inference form, per stack, HIP events over blocks of back-to-back launches.
usage (GPU box): [PPO_AMD_LIB=...variant.so] python tools/bf16x3_train_speed.py [stack ...]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_amd import _lib, models  # noqa: E402

lib = _lib.load()
torch.manual_seed(0)
net = models.DualHeadNet("impala", (4, 84, 84), 6, hidden_units=256, head_scale=0.1, head_bias=True, device="cuda")
st = _lib.current_stream()
stacks = [int(a) for a in sys.argv[1:]] or [0, 1, 2]
for stack in stacks:
    ch, hw = {0: (16, 42), 1: (32, 21), 2: (32, 11)}[stack]
    fwd_names = [f"encoder.stacks.{stack}.blocks.{bi}.conv{ci}" for bi in range(2) for ci in range(2)]
    bwd_names = [f"encoder.stacks.{stack}.blocks.{bi}.conv{ci}" for bi in (1, 0) for ci in (1, 0)]
    nbytes = int(lib.ppo_impala_stack_tail_bf16x3_packed_bytes())
    pk, pk_t = (torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(2))
    for names, buf, tr in ((fwd_names, pk, 0), (bwd_names, pk_t, 1)):
        wp = (ctypes.c_void_p * 4)(*[net.params[n + ".weight"].data_ptr() for n in names])
        _lib.check(lib.ppo_impala_stack_tail_pack_bf16x3(wp, buf.data_ptr(), ch, tr, st), "pack")
    bp = (ctypes.c_void_p * 4)(*[net.params[n + ".bias"].data_ptr() for n in fwd_names])
    for B in (128, 256):
        p = torch.randn(B, ch, hw, hw, device="cuda")
        g = torch.randn(B, ch, hw, hw, device="cuda")
        outs = [torch.empty_like(p) for _ in range(4)]
        gouts = [torch.empty_like(p) for _ in range(4)]
        masks = (ctypes.c_void_p * 4)(outs[2].data_ptr(), outs[1].data_ptr(), outs[0].data_ptr(), p.data_ptr())

        def infer():
            lib.ppo_impala_stack_tail_forward_bf16x3(p.data_ptr(), pk.data_ptr(), bp, None, None, None, outs[3].data_ptr(), B, ch, hw, hw, st)

        def fwd():
            lib.ppo_impala_stack_tail_forward_bf16x3(p.data_ptr(), pk.data_ptr(), bp, *[t.data_ptr() for t in outs], B, ch, hw, hw, st)

        def bwd():
            lib.ppo_impala_stack_tail_backward_bf16x3(g.data_ptr(), pk_t.data_ptr(), masks, *[t.data_ptr() for t in gouts], B, ch, hw, hw, st)
        res = {}
        for name, fn in (("infer", infer), ("fwd", fwd), ("bwd", bwd)):
            for _ in range(5):
                fn()
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
        mb = B * ch * hw * hw * 4 / 1e6
        print(f"stack {stack} ({ch}ch {hw}x{hw}) B={B}: inference {res['infer']:6.1f} us   training forward {res['fwd']:6.1f} us "
              f"({5 * mb / res['fwd']:.2f} TB/s of 5 maps)   backward-data {res['bwd']:6.1f} us ({9 * mb / res['bwd']:.2f} TB/s of 9 maps)")
