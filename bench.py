#!/usr/bin/env python3
"""bench.py — driver-facing benchmark of the MI355X-native PPO hot path.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (for N>1 launched by torch.distributed.run; RCCL only for
the barrier and the max-over-ranks time: env columns are independent, so the
data path has no collective — "scaling": "weak").  Prints ONE JSON line on
rank 0.  See DESIGN.md §Measurement for how every figure is defined.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md: 8.0 TB/s spec)
BYTES_PER_ELEM = 17     # fused adv+returns scan: read r4+v4+done1, write adv4+ret4


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--scan-envs", type=int, default=1 << 20, help="A of the bandwidth-regime scan")
    p.add_argument("--n-steps", type=int, default=256, help="rollout length N")
    p.add_argument("--no-cpu-baseline", action="store_true")
    return p.parse_args()


def init_dist(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path)")
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    return world, rank, local


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world):
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def scan_inputs(N, A, seed):
    """BASELINE.md §3 inputs: r~N(0,1), V~N(0,1), done~Bernoulli(0.01), generated in HBM."""
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(seed)
    r = torch.randn(N, A, generator=g, device=dev)
    v = torch.randn(N + 1, A, generator=g, device=dev)
    d = torch.rand(N, A, generator=g, device=dev) < 0.01
    return r, v, d


def run_scan(lib, r, v, d, adv, ret, N, A, gamma, lam_a, lam_r, stream):
    from ppo_amd import _lib
    rc = lib.ppo_gae_scan_f32(r.data_ptr(), v.data_ptr(), v[N].data_ptr(), d.data_ptr(), _lib.PPO_TERM_U8,
                              adv.data_ptr(), ret.data_ptr(), N, A, A, gamma, lam_a, lam_r, _lib.PPO_SCAN_AUTO,
                              stream)
    _lib.check(rc, "ppo_gae_scan_f32")


def cpu_baseline(N):
    """The oracle (scalar C port of rl/returns.py, 1 thread) on a bounded sample of the
    same workload: N x 65536 columns, repeated for ~10 s."""
    from oracle import returns as O
    A = 65536
    rng = np.random.default_rng(0)
    r = rng.normal(size=(N, A)).astype(np.float32)
    v = rng.normal(size=(N + 1, A)).astype(np.float32)
    d = rng.random((N, A)) < 0.01
    O.gae_and_returns(r, v[:N], v[N], d, 0.999, 0.95, 0.95)  # warm-up (+ builds the oracle)
    best, total, reps = 1e30, 0.0, 0
    while total < 10.0 and reps < 50:
        t0 = time.perf_counter()
        O.gae_and_returns(r, v[:N], v[N], d, 0.999, 0.95, 0.95)
        dt = time.perf_counter() - t0
        best = min(best, dt)
        total += dt
        reps += 1
    return {"value": round(BYTES_PER_ELEM * N * A / best / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"oracle/returns_oracle.c gae+td_lambda, N={N} A={A} bool terminals, best of {reps}",
            "melem_per_s": round(N * A / best / 1e6, 1), "host_cores_available": os.cpu_count()}


def main():
    args = parse()
    world, rank, local = init_dist(args)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    from ppo_amd import _lib
    lib = _lib.load()
    N, A = args.n_steps, args.scan_envs
    gamma, lam = 0.999, 0.95  # rl/config.py:769-773
    r, v, d = scan_inputs(N, A, seed=rank)
    adv = torch.empty_like(r)
    ret = torch.empty_like(r)
    stream = torch.cuda.current_stream().cuda_stream

    for _ in range(args.warmup):
        run_scan(lib, r, v, d, adv, ret, N, A, gamma, lam, lam, stream)
    barrier(world)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for k in range(args.steps):
        run_scan(lib, r, v, d, adv, ret, N, A, gamma, lam, lam, stream)
        ev[k + 1].record()
    barrier(world)
    wall = time.perf_counter() - t0
    wall = max_over_ranks(wall, world)
    kern_ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(args.steps)]
    kern_avg_s = sum(kern_ms) / len(kern_ms) / 1e3

    # parity gate printed with the number: sampled columns vs the oracle (columns are independent)
    parity = None
    if rank == 0:
        from oracle import returns as O
        cols = torch.arange(0, A, max(1, A // 257), device="cuda")[:257]
        oa, orr = O.gae_and_returns(r[:, cols].cpu().numpy(), v[:N][:, cols].cpu().numpy(),
                                    v[N][cols].cpu().numpy(), d[:, cols].cpu().numpy(), gamma, lam, lam)
        parity = bool(np.array_equal(adv[:, cols].cpu().numpy(), oa) and np.array_equal(ret[:, cols].cpu().numpy(), orr))

    bytes_per_launch = BYTES_PER_ELEM * N * A
    achieved = bytes_per_launch / kern_avg_s / 1e9
    out = {
        "metric": "GAE-scan HBM GB/s",
        "value": round(world * bytes_per_launch * args.steps / wall / 1e9, 2),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64 carry / f32 io",
        "data": "synthetic",
        "config": {"workload": f"fused GAE+lambda-returns scan, N={N}, A={A} per GPU, bool terminals, "
                               f"gamma={gamma} lambda={lam}", "regime": "columns" if A > 65536 else "tiles"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                     "kernel": "gae_columns_kernel" if A > 65536 else "gae_tiles_kernel",
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "avg_kernel_ms": round(kern_avg_s * 1e3, 4)},
        "parity_bit_exact_vs_oracle": parity,
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
