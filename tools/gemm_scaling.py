"""ppo_gemm_f32 time against K (dense dW: K = batch) and against the grid size (dense dX: K = 256): the slope is
~1.85 us per 32-deep slab and the dX time grows linearly with the workgroup count at every occupancy, i.e. the 64x64
tiles (16 FLOP per byte of operand traffic) are bound by L2 / fabric bandwidth (~3 TB/s of re-reads), not by MFMA.
Usage: python tools/gemm_scaling.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_amd import _lib
lib = _lib.load()
dev = torch.device("cuda")
F, H = 3872, 256
def p(t): return None if t is None else t.data_ptr()
def run(fn, reps=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for B in (64, 128, 256, 512, 1024, 2048):
    x = torch.randn(B, F, device=dev); dh = torch.randn(B, H, device=dev); gw = torch.empty(H, F, device=dev); gx = torch.empty(B, F, device=dev)
    w = torch.randn(H, F, device=dev)
    t_dw = run(lambda: lib.ppo_gemm_f32(p(dh), 1, H, 0, p(x), F, 1, 1, None, None, p(gw), F, H, F, B, None, 0, _lib.current_stream()))
    t_dx = run(lambda: lib.ppo_gemm_f32(p(dh), H, 1, 0, p(w), F, 1, 0, None, p(x), p(gx), F, B, F, H, None, 0, _lib.current_stream()))
    print(f"B={B:5d}: dW (K=B, 244 WGs) {t_dw:7.1f} us | dX (K=256, {((B+63)//64)*61} WGs) {t_dx:7.1f} us")
