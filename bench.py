#!/usr/bin/env python3
"""bench.py — driver-facing benchmark of the MI355X-native PPO hot path.

    python bench.py --gpus N --steps K --warmup W

A "step" is ONE PPO iteration of BASELINE.json's config 2 on synthetic data: 256 envs per GPU x 256
steps of 84x84x4 uint8 observations (rollout with the policy forward + on-device sampling), the fused
GAE/lambda-return scan, and 2 policy epochs of 256-sample minibatches (forward, PPO loss, backward,
global-norm clip, Adam) through the hand-written HIP IMPALA network.  value = env-steps/s summed over
ranks (the reference's IPS, rl/ppo.py:354-365).  One process per GPU; for N > 1 the envs are sharded
(weak scaling: 256 envs per rank, global minibatch 256*N so the optimiser-step count is unchanged)
and the only collectives are the RCCL gradient all-reduce per optimiser step and the advantage
moments per batch.

Also reported on the same JSON line:
  roofline      the kernel with the largest share of the step (a conv weight-gradient launch), timed
                live with HIP events on the launch stream during the timed region, against the fp32
                MFMA peak;
  gae_scan      the fused GAE scan at the bandwidth-regime size (N=256, A=2^20) against HBM peak, and
                its latency at the config size;
  cpu_baseline  the same PPO iteration through oracle/model_torch.py (plain torch CPU operators — what
                the reference runs with --device=cpu) on a bounded sample, extrapolated to env-steps/s.
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

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_* dense peak (MI355X_MICROARCH.md, Matrix cores)
PROBE_TRAFFIC_BYTES = int((2 * 11295 + 15009) * 1024)  # conv3x3_kernel<32,32,21,21,IN_RELU> forward, batch 256
STACK_CHAIN_TRAFFIC_BYTES = int((2 * 8625.0 + 76879.8) * 1024)  # stack_full_kernel chained training forward, batch 256
STACK_TAIL_TRAFFIC_BYTES = int((2 * 7826.4 + 60447.7) * 1024)  # stack_tail_kernel<32,21,21> training forward, batch 256
SCAN_BYTES_PER_ELEM = 17      # fused adv+returns scan: read r4+v4+done1, write adv4+ret4
FWD_MFLOP_PER_SAMPLE = 108.4  # IMPALA forward at 4x84x84


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--agents", type=int, default=256, help="envs per GPU")
    p.add_argument("--n-steps", type=int, default=256, help="rollout length N")
    p.add_argument("--scan-envs", type=int, default=1 << 20, help="A of the bandwidth-regime scan")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-scan", action="store_true")
    p.add_argument("--scan-only", action="store_true",
                   help="only the gae_scan section (used for the rocprofv3 --pmc passes: counter collection "
                        "around the full PPO iteration segfaults inside rocprofv3 on this pool)")
    return p.parse_args()


def init_dist():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path)")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    return world, rank, local


def barrier(world):
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world):
    if world == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device="cuda")
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


# ----------------------------------------------------------------------------- live kernel probe
class CallProbe:
    """Wraps DualHeadNet._call so that every launch of one C-ABI entry point with one geometry is
    bracketed by HIP events on the launch stream (torch's current stream)."""

    def __init__(self, net, fn_name, match):
        self.net, self.fn_name, self.match = net, fn_name, match
        self.events = []
        self.seen = None  # the entry point the timed launches went through
        self.enabled = False
        self._orig = net._call
        net._call = self._call

    def _call(self, fn_name, *a):
        if self.enabled and fn_name in self.fn_name and self.match(fn_name, a):
            self.seen = fn_name
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._orig(fn_name, *a)
            e1.record()
            self.events.append((e0, e1))
        else:
            self._orig(fn_name, *a)

    def avg_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.events) / max(1, len(self.events))


# ----------------------------------------------------------------------------- GAE scan section
def bench_scan(lib, N, A_big, A_cfg):
    from ppo_amd import _lib
    out = {}
    for tag, A, reps in (("bandwidth", A_big, 10), ("config", A_cfg, 50)):
        dev = torch.device("cuda")
        g = torch.Generator(device=dev).manual_seed(0)
        r = torch.randn(N, A, generator=g, device=dev)
        v = torch.randn(N + 1, A, generator=g, device=dev)
        d = torch.rand(N, A, generator=g, device=dev) < 0.01
        adv, ret = torch.empty_like(r), torch.empty_like(r)

        def go():
            rc = lib.ppo_gae_scan_f32(r.data_ptr(), v.data_ptr(), v[N].data_ptr(), d.data_ptr(), _lib.PPO_TERM_U8,
                                      adv.data_ptr(), ret.data_ptr(), N, A, A, 0.999, 0.95, 0.95, _lib.PPO_SCAN_AUTO,
                                      torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, "ppo_gae_scan_f32")
        for _ in range(3):
            go()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for k in range(reps):
            go()
            ev[k + 1].record()
        torch.cuda.synchronize()
        per = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(reps))
        ms = sum(per) / reps
        gbps = SCAN_BYTES_PER_ELEM * N * A / (ms * 1e-3) / 1e9
        out[tag] = {"N": N, "A": A, "avg_kernel_us": round(ms * 1e3, 2), "achieved_GBps": round(gbps, 1),
                    "min_kernel_us": round(per[0] * 1e3, 2), "max_kernel_us": round(per[-1] * 1e3, 2),
                    "launches_timed": reps,
                    "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBPS, 4),
                    "kernel": "gae_columns_kernel" if A > 65536 else "gae_tiles_kernel"}
        if tag == "bandwidth":
            from oracle import returns as O
            cols = torch.arange(0, A, max(1, A // 129), device=dev)[:129]
            oa, orr = O.gae_and_returns(r[:, cols].cpu().numpy(), v[:N][:, cols].cpu().numpy(), v[N][cols].cpu().numpy(),
                                        d[:, cols].cpu().numpy(), 0.999, 0.95, 0.95)
            out[tag]["bit_exact_vs_oracle"] = bool(np.array_equal(adv[:, cols].cpu().numpy(), oa) and
                                                   np.array_equal(ret[:, cols].cpu().numpy(), orr))
    out["peak_GBps"] = HBM_PEAK_GBPS
    out["algorithmic_bytes_per_element"] = SCAN_BYTES_PER_ELEM
    return out


# ----------------------------------------------------------------------------- CPU baseline
def cpu_baseline(N, A, epochs, mb):
    """The PPO iteration on the host through plain torch CPU ops (oracle/model_torch.py), bounded:
    time one rollout forward of `fb` observations and one train minibatch of `tb` samples, then
    extrapolate to a full iteration of N*A env steps."""
    from oracle import model_torch as R, returns as O
    from ppo_amd.models import ImpalaSpec, init_impala_parameters
    threads = torch.get_num_threads()
    torch.manual_seed(1)
    init = init_impala_parameters(ImpalaSpec((4, 84, 84)), 6, 1, 0.1, True)
    fb, tb = 64, 64
    rng = np.random.default_rng(0)
    xf = torch.from_numpy(rng.integers(0, 256, (fb, 4, 84, 84), dtype=np.uint8))
    xt = torch.from_numpy(rng.integers(0, 256, (tb, 4, 84, 84), dtype=np.uint8))
    actions = torch.from_numpy(rng.integers(0, 6, (tb,)))
    lpac = torch.full((tb,), -1.79)
    adv = torch.from_numpy(rng.normal(size=(tb,)).astype(np.float32))
    ret = torch.from_numpy(rng.normal(size=(tb, 1)).astype(np.float32))
    sd = {k: v.clone().requires_grad_(True) for k, v in init.items()}
    opt = torch.optim.Adam([p for p in sd.values()], lr=2.5e-4, eps=1e-5)

    def fwd():
        with torch.no_grad():
            R.forward(init, xf.float() / 255.0)

    def train():
        opt.zero_grad(set_to_none=True)
        R.ppo_loss(R.forward(sd, xt.float() / 255.0), actions, lpac, adv, ret).backward()
        torch.nn.utils.clip_grad_norm_([p for p in sd.values() if p.grad is not None], 20.0)
        opt.step()
    times = {}
    for name, fn in (("fwd", fwd), ("train", train)):
        fn()
        t0, reps = time.perf_counter(), 0
        while time.perf_counter() - t0 < 6.0 and reps < 20:
            fn()
            reps += 1
        times[name] = (time.perf_counter() - t0) / reps
    r = rng.normal(size=(N, A)).astype(np.float32)
    v = rng.normal(size=(N + 1, A)).astype(np.float32)
    d = rng.random((N, A)) < 0.01
    t0 = time.perf_counter()
    O.gae_and_returns(r, v[:N], v[N], d, 0.999, 0.95, 0.95)
    t_scan = time.perf_counter() - t0
    per_iter = (N + 1) * A * times["fwd"] / fb + epochs * N * A * times["train"] / tb + t_scan
    return {"value": round(N * A / per_iter, 1), "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"torch-CPU restatement (oracle/model_torch.py): 1 rollout forward of {fb} obs "
                      f"({times['fwd']*1e3:.0f} ms) + 1 PPO minibatch of {tb} ({times['train']*1e3:.0f} ms) + C-oracle GAE "
                      f"scan {N}x{A} ({t_scan*1e3:.1f} ms), extrapolated to one iteration of {N*A} env steps",
            "host_cores_available": os.cpu_count()}


# ----------------------------------------------------------------------------- main
def main():
    a = parse()
    world, rank, local = init_dist()
    from ppo_amd import _lib, envs, logger, models, rollout
    from ppo_amd.config import args
    lib = _lib.load()
    N, A = a.n_steps, a.agents
    if a.scan_only:
        print(json.dumps({"gae_scan": bench_scan(lib, N, a.scan_envs, A)}), flush=True)
        return
    mb = 256
    args.setup([f"--agents={A}", f"--n_steps={N}", "--model_architecture=single", "--model_encoder=impala",
                "--env_type=synthetic", "--env_embed_time=False", "--seed=1", f"--device=cuda:{local}",
                f"--policy_opt_mini_batch_size={mb * world}", "--policy_opt_epochs=2", "--disable_logging=True",
                "--upload_batch=True", "--env_reward_normalization=off"])
    torch.manual_seed(1)
    np.random.seed(1 + rank)
    obs_shape, n_actions = envs.get_env_spec()
    model = models.TVFModel(encoder="impala", input_dims=obs_shape, actions=n_actions, device=f"cuda:{local}",
                            architecture="single", hidden_units=args.model.hidden_units, head_scale=args.model.head_scale,
                            head_bias=args.model.head_bias)
    runner = rollout.Runner(model, logger.Logger(quiet=True))
    runner.vec_env = envs.create_envs_classic(rank=rank, world=world)
    runner.reset()
    # dominant kernel of the step (profiles/r01e: largest total time): the 32->32 21x21 forward convolution of the
    # residual blocks, conv3x3_kernel<32,32,21,21,..,IN_RELU>; the train-minibatch launches (n == mb) are timed
    # of the residual-block convolutions of the 21x21 stack: one fused launch per training forward
    # (stack_tail_kernel<32,21,21>: 4 convolutions, image resident in LDS) or, with PPO_AMD_FUSE_STACK_TAIL=0, the
    # individual conv3x3_kernel<32,32,21,21,IN_RELU> launches.  Only minibatch-sized training launches are timed.
    def probe_match(fn_name, c):
        if fn_name == "ppo_impala_stack_chain_forward_f32":  # (in, pre_w, pre_b, pre_a0..pre_q1, w, b, pooled, argmax,
            return c[3] is not None and (c[15], c[16], c[17], c[18]) == (mb, 32, 21, 21)  # a0..q1, n, channels, h, w)
        if fn_name == "ppo_impala_stack_tail_forward_f32":  # (in, w[4], b[4], a0, q0, a1, q1, n, channels, h, w)
            return c[3] is not None and (c[7], c[8], c[9], c[10]) == (mb, 32, 21, 21)
        return c[1] == 1 and (c[6], c[7], c[8], c[9], c[10]) == (mb, 32, 32, 21, 21)

    probe = CallProbe(model.policy_net, ("ppo_conv3x3_forward_f32", "ppo_conv3x3_forward_packed_f32",
                                         "ppo_impala_stack_tail_forward_f32", "ppo_impala_stack_chain_forward_f32"),
                      probe_match)
    conv_flops = 2 * 9 * 32 * 32 * 21 * 21 * mb

    def iteration():
        runner.generate_rollout()
        runner.calculate_returns()
        runner.train()

    for _ in range(a.warmup):
        iteration()
    barrier(world)
    probe.enabled = True
    phase = {"rollout": 0.0, "returns": 0.0, "train": 0.0}
    t0 = time.perf_counter()
    for _ in range(a.steps):
        t = time.perf_counter()
        runner.generate_rollout()
        phase["rollout"] += time.perf_counter() - t
        t = time.perf_counter()
        runner.calculate_returns()
        phase["returns"] += time.perf_counter() - t
        t = time.perf_counter()
        runner.train()
        phase["train"] += time.perf_counter() - t
    barrier(world)
    wall = max_over_ranks(time.perf_counter() - t0, world)
    probe.enabled = False
    stats = runner.fetch_stats()

    env_steps = world * N * A * a.steps
    if not probe.events:
        raise SystemExit("bench.py: the roofline probe saw no launch of its kernel (entry point renamed?)")
    kern_ms = probe.avg_ms()
    fused = probe.seen == "ppo_impala_stack_tail_forward_f32"
    map_bytes = 32 * 21 * 21 * 4 * mb
    small_bytes = 32 * 11 * 11 * 4 * mb
    if probe.seen == "ppo_impala_stack_chain_forward_f32":
        # the 21x21 stack's 4 block convolutions, the 11x11 stack's first convolution (on the 21x21 map) + max-pool and
        # its 4 block convolutions; reads the 21x21 pooled map, writes the 4 + 4 maps the backward pass needs, the
        # pooled 11x11 map and its uint8 argmax
        probe_flops = 5 * conv_flops + 4 * (2 * 9 * 32 * 32 * 11 * 11 * mb)
        probe_bytes = 5 * map_bytes + 5 * small_bytes + small_bytes // 4
        probe_traffic = STACK_CHAIN_TRAFFIC_BYTES
        probe_kernel = ("stack_full_kernel<32,21,21 -> 11,11> chained training forward: residual blocks of the 21x21 stack + "
                        "the whole 11x11 stack (9 convolutions + max-pool) in one launch, maps resident in LDS "
                        "(ppo_impala_stack_chain_forward_f32, minibatch launches)")
        probe_source = "profiles/r01l_stack_tail_hbm_traffic.md"
    elif fused:   # reads the block input, writes a0, q0, a1, q1 for the backward pass
        probe_flops, probe_bytes, probe_traffic = 4 * conv_flops, 5 * map_bytes, STACK_TAIL_TRAFFIC_BYTES
        probe_kernel = ("stack_tail_kernel<32,21,21> training forward: the 4 residual-block convolutions of the 21x21 "
                        "stack in one launch (ppo_impala_stack_tail_forward_f32, minibatch launches)")
        probe_source = "profiles/r01l_stack_tail_hbm_traffic.md"
    else:
        probe_flops, probe_bytes, probe_traffic = conv_flops, 2 * map_bytes, PROBE_TRAFFIC_BYTES
        probe_kernel = "conv3x3_kernel<32,32,21,21,IN_RELU> forward (ppo_conv3x3_forward_f32, minibatch launches)"
        probe_source = "profiles/r01j_conv_hbm_traffic.md"
    tflops = probe_flops / (kern_ms * 1e-3) / 1e12
    samples_fwd = (N + 1) * A + args.policy_opt.epochs * N * A
    samples_bwd = args.policy_opt.epochs * N * A
    model_tflops = (samples_fwd + 2 * samples_bwd) * FWD_MFLOP_PER_SAMPLE * 1e6 * a.steps / wall / 1e12
    out = {
        "metric": "env-steps/sec (Pong-shaped synthetic, 256 envs/GPU)",
        "value": round(env_steps / wall, 1),
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(wall / a.steps * 1e3, 2),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"PPO iteration: {A} envs/GPU x {N} steps, obs {obs_shape} uint8, IMPALA-CNN single "
                               f"architecture ({model.model_size()} params), {args.policy_opt.epochs} policy epochs, "
                               f"global minibatch {mb * world}, Adam, synthetic env (uniform uint8 obs, N(0,1) reward, "
                               f"p_done 0.01)", "envs_per_gpu": A, "n_steps": N, "global_minibatch": mb * world,
                   "parallelism": f"dp{world}"},
        "roofline": {"bound": "mfma", "achieved": round(tflops, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tflops / MFMA_F32_PEAK_TFLOPS, 4),
                     # HBM bytes per launch of this kernel at this shape: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
                     # separate passes over tools/conv_tune (counter collection around the whole PPO iteration
                     # segfaults in rocprofv3 on this pool), 2 x FETCH_SIZE + WRITE_SIZE KiB as the guide prescribes
                     # for gfx950; algorithmic = input + output = 28.9 MB, the 8/6 halo-row re-read accounts for the rest
                     "traffic": probe_traffic if mb == 256 else None,
                     "traffic_source": probe_source,
                     "algorithmic_bytes_per_launch": probe_bytes,
                     "kernel": probe_kernel,
                     "algorithmic_flops_per_launch": probe_flops, "avg_kernel_ms": round(kern_ms, 4),
                     "launches_timed": len(probe.events)},
        "phase_seconds_per_step": {k: round(v / a.steps, 4) for k, v in phase.items()},
        "model_tflops_whole_step": round(model_tflops / world, 2),
        "train_stats": {k: round(float(v), 6) for k, v in stats.items()},
    }
    if rank == 0:
        if world == 1 and not a.no_scan:
            out["gae_scan"] = bench_scan(lib, N, a.scan_envs, A)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, A, args.policy_opt.epochs, mb)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
