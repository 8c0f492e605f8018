#!/usr/bin/env python3
"""Microbench: GAE scan kernel time over (A, regime); prints a table. GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd import _lib

lib = _lib.load()
N = 256
dev = torch.device("cuda")
print(f"{'A':>9} {'regime':>8} {'us':>10} {'GB/s(17B)':>10}")
for A in [256, 512, 1024, 4096, 16384, 32768, 65536, 262144, 1 << 20, 1 << 22]:
    g = torch.Generator(device=dev).manual_seed(0)
    r = torch.randn(N, A, generator=g, device=dev)
    v = torch.randn(N + 1, A, generator=g, device=dev)
    d = torch.rand(N, A, generator=g, device=dev) < 0.01
    adv = torch.empty_like(r); ret = torch.empty_like(r)
    st = torch.cuda.current_stream().cuda_stream
    for regime, name in ((1, "columns"), (2, "tiles")):
        def go():
            rc = lib.ppo_gae_scan_f32(r.data_ptr(), v.data_ptr(), v[N].data_ptr(), d.data_ptr(), 1, adv.data_ptr(),
                                      ret.data_ptr(), N, A, A, 0.999, 0.95, 0.95, regime, st)
            assert rc == 0
        for _ in range(3): go()
        torch.cuda.synchronize()
        reps = 20
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): go()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"{A:>9} {name:>8} {us:>10.2f} {17*N*A/us/1e3:>10.1f}", flush=True)
