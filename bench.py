#!/usr/bin/env python3
"""bench.py — driver-facing benchmark of the MI355X-native PPO hot path.

    python bench.py --gpus N --steps K --warmup W

A "step" is ONE PPO iteration of BASELINE.json's config 2 on synthetic data: 256 envs per GPU x 256
steps of 84x84x4 uint8 observations (rollout with the policy forward + on-device sampling), the fused
GAE/lambda-return scan, and 2 policy epochs of 256-sample minibatches (forward, PPO loss, backward,
global-norm clip, Adam) through the hand-written HIP IMPALA network.  value = env-steps/s summed over
ranks (the reference's IPS, rl/ppo.py:354-365).  One process per GPU; for N > 1 the envs are sharded
(weak scaling: 256 envs per rank, global minibatch 256*N so the optimiser-step count is unchanged)
and the collectives are the RCCL gradient all-reduce per optimiser step (two buckets, the large one
overlapped with the convolution backward) and the advantage moments per batch.

Launch forms: `python bench.py --gpus N` starts its own N rank processes (children, started before this
process touches the GPU) and relays rank 0's JSON line; under `python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N` (WORLD_SIZE already set) it is one of the ranks.  `--gpus` that
disagrees with WORLD_SIZE, or asks for more GPUs than the node has, is an error, never a silent N=1.

Also reported on the same JSON line:
  roofline            the kernel with the largest share of the step's GPU time (the chained LDS-resident
                      stack launch of the training forward), timed live with HIP events on its launch
                      stream during the timed region, against the fp32 MFMA peak;
  roofline_by_kernel  the time-weighted picture: every kernel class of one extra (untimed) iteration of the
                      same run, bracketed by HIP events on its launch stream — share of GPU kernel time,
                      achieved TFLOP/s or GB/s from the algorithmic FLOPs / bytes, fraction of the peak that
                      bounds it;
  gae_scan            the fused GAE scan at the bandwidth-regime size (N=256, A=2^20) against HBM peak, and
                      its latency at the config size;
  cpu_baseline        the same PPO iteration through oracle/model_torch.py (plain torch CPU operators — what
                      the reference runs with --device=cpu) on a bounded sample (one 256-observation rollout
                      forward, one 256-sample minibatch), extrapolated to env-steps/s.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_* dense peak (MI355X_MICROARCH.md, Matrix cores)
# the opt-in split-bf16 launches (--precision low|medium): three bf16 MFMAs per product, so their algorithmic FLOP/s are
# bounded by a third of the dense bf16 MFMA peak (~2.5 PFLOP/s: v_mfma_f32_16x16x32_bf16, 16 cycles per 16x16x32)
MFMA_BF16X3_PEAK_TFLOPS = 2516.6 / 3
SCAN_BYTES_PER_ELEM = 17      # fused adv+returns scan: read r4+v4+done1, write adv4+ret4
FWD_MFLOP_PER_SAMPLE = 108.4  # IMPALA forward at 4x84x84
# HBM bytes per launch of the roofline kernel from rocprofv3 --pmc passes over THIS command (tools/pmc_bench.sh
# -> profiles/<tag>_bench_hbm_traffic.json); absent file => traffic null
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "bench_hbm_traffic.json")  # the latest pass (tools/round_artifacts.py)


# BASELINE.json `configs`, as synthetic workloads of the named shape (synthetic rollouts; the simulators
# are not installed).  `pong` is the configuration the headline metric is quoted on and the default line.
CONFIGS = {
    "pong": dict(agents=256, obs=(4, 84, 84), actions=6, dist="discrete", metric="env-steps/sec (Pong-shaped synthetic, 256 envs/GPU)",
                 flags=["--model_architecture=single", "--model_encoder=impala", "--env_type=synthetic",
                        "--env_embed_time=False", "--policy_opt_epochs=2"]),
    "procgen": dict(agents=1024, obs=(3, 64, 64), actions=15, dist="discrete",
                    metric="env-steps/sec (Procgen-shaped synthetic, 1024 envs/GPU)",
                    flags=["--model_architecture=single", "--model_encoder=impala", "--env_type=synthetic",
                           "--env_synthetic_shape=3,64,64", "--env_synthetic_actions=15", "--env_embed_time=False",
                           "--policy_opt_epochs=2"]),
    "humanoid_tvf": dict(agents=256, obs=(377,), actions=17, dist="gaussian",
                         metric="env-steps/sec (Humanoid-shaped synthetic, 256 envs/GPU, MLP + TVF)",
                         flags=["--model_architecture=dual", "--model_encoder=mlp", "--model_hidden_units=256",
                                "--env_type=mujoco", "--env_name=Humanoid", "--tvf_enabled=True", "--tvf_value_heads=128",
                                "--tvf_max_horizon=30000", "--value_opt_mini_batch_size=256",
                                "--distil_opt_mini_batch_size=256"]),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--config", default="pong", choices=sorted(CONFIGS),
                   help="workload: pong = BASELINE.json configs[1] (the driver's line, default); procgen = configs[2]; "
                        "humanoid_tvf = configs[4]")
    p.add_argument("--agents", type=int, default=None, help="envs per GPU (default: the config's)")
    p.add_argument("--n-steps", type=int, default=256, help="rollout length N")
    p.add_argument("--scan-envs", type=int, default=1 << 20, help="A of the bandwidth-regime scan")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-scan", action="store_true")
    p.add_argument("--no-kernel-table", action="store_true", help="skip the extra iteration behind roofline_by_kernel")
    p.add_argument("--scan-only", action="store_true", help="only the gae_scan section (rocprofv3 --pmc passes of the scan)")
    p.add_argument("--tvf-only", action="store_true", help="only the tvf_returns section (timing / rocprofv3 passes)")
    p.add_argument("--tvf-heads", type=int, default=108, help="K = V of the tvf_returns section")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend of the ranks (nccl = RCCL)")
    p.add_argument("--precision", default="high", choices=["low", "medium", "high"],
                   help="high (default, THE benchmark): exact float32 everywhere.  low / medium: the reference's flag "
                        "(train.py:166-178 allows TF32 there) - the 32-channel residual blocks run as split-bf16 launches "
                        "(3 bf16 MFMAs per product); the line's dtype says so and it is never the headline")
    p.add_argument("--no-opt-in", action="store_true",
                   help="skip the extra leg that times the same workload under --precision=medium (N = 1, IMPALA configs, "
                        "reported beside the line as opt_in_precision; never `value`)")
    p.add_argument("--no-affinity", action="store_true",
                   help="do not pin the rank (and its env threads) to the cores of its GPU's NUMA node (ppo_amd/affinity.py)")
    p.add_argument("--wall-limit", type=float, default=1500.0,
                   help="launcher: seconds after which still-running ranks are stopped and the run fails")
    p.add_argument("--dist-timeout", type=float, default=180.0, help="timeout of the process group's collectives, seconds")
    p.add_argument("--fail-rank", type=int, default=-1, help="launcher test hook: this rank exits non-zero at start-up")
    p.add_argument("--hang-rank", type=int, default=-1, help="launcher test hook: this rank sleeps instead of joining")
    p.add_argument("--rendezvous-only", action="store_true",
                   help="start the ranks, form the process group, all-reduce one number, print who was seen; no GPU work "
                        "(the CPU test of the launcher uses it with --backend gloo)")
    return p.parse_args()


# ----------------------------------------------------------------------------- rank launcher
def count_gpus_without_hip():
    """GPUs this process could address, read from the KFD topology in sysfs and the *_VISIBLE_DEVICES variables: no
    torch.cuda / HIP call, so the launcher stays a process that has never initialised the GPU BY CONSTRUCTION
    (torch.cuda.device_count() can fall back to hipGetDeviceCount).  None = sysfs unreadable: the ranks check."""
    nodes = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir(nodes):
        return 0
    have = 0
    try:
        for d in os.listdir(nodes):
            props = dict(ln.split()[:2] for ln in open(os.path.join(nodes, d, "properties")) if len(ln.split()) >= 2)
            have += int(props.get("simd_count", "0")) > 0
    except OSError:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            have = min(have, len([x for x in v.split(",") if x.strip() != ""]))
    return have


def launch_ranks(a):
    """`python bench.py --gpus N` with no WORLD_SIZE: start N fresh rank processes, watch ALL of them, relay rank 0's
    stdout.  The first rank that exits non-zero (bad device, OOM, port clash, a failed collective) or the wall limit
    ends the run: the remaining ranks are terminated (then killed), the failing rank's stderr tail is printed and
    the exit code is non-zero — a hang in one rank's RCCL call can no longer leave the bench waiting forever.  The
    children are ordinary children of a parent that never touches the GPU — never a re-exec of a GPU-holding process."""
    import signal
    import tempfile
    n = a.gpus
    if not a.rendezvous_only:
        have = count_gpus_without_hip()
        if have is not None and have < n:
            raise SystemExit(f"bench.py: --gpus {n} requested but this node exposes {have} GPU(s); refusing to report a "
                             f"{n}-GPU number from fewer devices")
    import socket
    with socket.socket() as s:  # a free rendezvous port unless the caller fixed one
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"),
               MASTER_PORT=os.environ.get("MASTER_PORT", str(port)), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs, outs, errs = [], [], []
    for r in range(n):
        outs.append(tempfile.TemporaryFile(mode="w+"))
        errs.append(tempfile.TemporaryFile(mode="w+"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]],
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=outs[r], stderr=errs[r],
                                      start_new_session=True))  # own process group: env threads / workers die with it

    def tail(f, nbytes=3000):
        f.flush()
        f.seek(0, os.SEEK_END)
        f.seek(max(0, f.tell() - nbytes))
        return f.read()

    def stop_all():
        for sig, grace in ((signal.SIGTERM, 5.0), (signal.SIGKILL, 5.0)):
            alive = [p for p in procs if p.poll() is None]
            if not alive:
                return
            for p in alive:
                try:
                    os.killpg(p.pid, sig)  # exactly the groups started above
                except ProcessLookupError:
                    pass
            t_end = time.monotonic() + grace
            while time.monotonic() < t_end and any(p.poll() is None for p in alive):
                time.sleep(0.05)

    deadline = time.monotonic() + a.wall_limit
    failed = None
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = f"rank(s) failed (rank, exit code): {bad}"
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                bad = [(r, None) for r, c in enumerate(codes) if c is None]
                failed = f"wall limit of {a.wall_limit:.0f} s exceeded; still running: ranks {[r for r, _ in bad]}"
                break
            time.sleep(0.05)
    finally:
        stop_all()
    outs[0].seek(0)
    sys.stdout.write(outs[0].read())
    sys.stdout.flush()
    if failed:
        for r, _c in bad:
            sys.stderr.write(f"---- bench.py rank {r} stderr (tail) ----\n{tail(errs[r])}\n")
        raise SystemExit(f"bench.py: {failed}; the other ranks were stopped")
    return 0


def init_dist(a):
    import datetime
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch {a.gpus} ranks "
                         f"(`python bench.py --gpus {a.gpus}` does it itself)")
    if a.fail_rank == rank:  # test hook of the launcher: this rank dies before the rendezvous
        raise SystemExit(f"bench.py: rank {rank} told to fail (--fail-rank)")
    if a.hang_rank == rank:
        time.sleep(3600)
    timeout = datetime.timedelta(seconds=a.dist_timeout)
    if a.rendezvous_only:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.distributed.init_process_group(a.backend, rank=rank, world_size=world, timeout=timeout)
        return world, rank, local
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path)")
    if local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} has LOCAL_RANK {local} but this node exposes "
                         f"{torch.cuda.device_count()} GPU(s)")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {"device_id": torch.device("cuda", local)} if a.backend == "nccl" else {}
        torch.distributed.init_process_group(a.backend, rank=rank, world_size=world, timeout=timeout, **kw)
        # first collective of the run, before any timed work: 4 bytes through the data-path backend (RCCL), then a
        # digest comparison; a rank that cannot reach its peers fails HERE, inside `timeout`, with a message
        from ppo_amd import parallel
        t = torch.ones(1, dtype=torch.float32, device="cuda")
        torch.distributed.all_reduce(t)
        torch.cuda.synchronize()
        seen = int(t.item())
        parallel.assert_identical_across_ranks([t], "the rendezvous all-reduce result")
        if seen != world:
            raise SystemExit(f"bench.py: rendezvous all-reduce saw {seen} ranks, expected {world}")
        if rank == 0:
            print(f"bench.py: {seen} ranks seen over {torch.distributed.get_backend()} before any timed work",
                  file=sys.stderr, flush=True)
    return world, rank, local


def barrier(world):
    import torch
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world):
    import torch
    if world == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device="cuda")
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


# ----------------------------------------------------------------------------- live kernel probes
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
        import torch
        if self.enabled and fn_name in self.fn_name and self.match(fn_name, a):
            self.seen = fn_name
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._orig(fn_name, *a)
            e1.record()
            self.events.append((e0, e1))
        else:
            self._orig(fn_name, *a)

    def remove(self):
        self.net._call = self._orig

    def avg_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.events) / max(1, len(self.events))


def kernel_source_hash():
    """sha256 (first 16 hex digits) over ppo_amd/csrc: what a PMC traffic record must have been measured on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ppo_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def forward_mflop(obs_shape, n_actions, hidden, config):
    """Algorithmic forward MFLOP per sample: 2 * 9 * Cin * Cout per output pixel of every convolution
    of the IMPALA encoder (3 stacks of 16 / 32 / 32 channels: first convolution at the input size, 3x3/s2 max-pool, two
    residual blocks of two convolutions) + the dense layer; MLP: the two dense layers + heads."""
    if len(obs_shape) == 1:
        heads = 2 * n_actions + 1 + (128 if config == "humanoid_tvf" else 0)
        return 2.0 * (obs_shape[0] * hidden + hidden * hidden + hidden * heads) / 1e6
    c, h, w = obs_shape
    total = 0.0
    for cout in (16, 32, 32):
        total += 2.0 * 9 * c * cout * h * w
        h, w = _half(h), _half(w)
        total += 4 * 2.0 * 9 * cout * cout * h * w
        c = cout
    total += 2.0 * c * h * w * hidden
    return total / 1e6


def _conv(n, cin, cout, h, w):
    return 2.0 * 9 * cin * cout * h * w * n


def _half(x):
    return (x + 1) // 2


# entry point -> (label, algorithmic FLOPs, algorithmic HBM bytes) of one launch, from the C-ABI arguments
# (include/ppo_amd.h).  FLOPs bound the MFMA kernels, bytes the streaming ones; neither => latency-bound helper.
def describe_call(fn, a, train_batch=256):
    label, flops, nbytes = _describe_call(fn, a)
    n = _batch_of(fn, a)
    if n is not None and n != train_batch:
        label += f" [n={n}]"  # rollout-geometry launches (half-batch env groups) listed beside the training ones
    return label, flops, nbytes


def _batch_of(fn, a):
    if fn in ("ppo_conv3x3_forward_f32", "ppo_conv3x3_forward_packed_f32", "ppo_conv3x3_pool_forward_f32",
              "ppo_conv3x3_pool_forward_packed_f32"):
        return a[6]
    if fn == "ppo_impala_stack_tail_forward_f32":
        return a[7]
    if fn == "ppo_impala_stack_full_forward_f32":
        return a[9]
    if fn == "ppo_impala_stack_chain_forward_f32":
        return a[15]
    if fn == "ppo_impala_stack16_forward_f32":
        return a[-4]
    if fn == "ppo_impala_stack_chain_split_forward_f32":
        return a[8]
    if fn in ("ppo_impala_stack_tail_forward_bf16x3", "ppo_impala_stack_tail_backward_bf16x3",
              "ppo_impala_stack_tail_backward_signs_bf16x3"):
        return a[7]
    if fn == "ppo_impala_stack_tail_forward_signs_bf16x3":
        return a[8]
    if fn == "ppo_conv3x3_bf16x3":
        return a[5]
    if fn == "ppo_conv3x3_pool_bf16x3":
        return a[6]
    if fn in ("ppo_dense_heads_forward_f32", "ppo_dense_heads_act_forward_f32", "ppo_dense_heads_loss_forward_f32"):
        return a[9]
    if fn == "ppo_conv3x3_block_forward_packed_f32":
        return a[6]
    return None


def _describe_call(fn, a):
    f32 = 4
    if fn in ("ppo_conv3x3_forward_f32", "ppo_conv3x3_forward_packed_f32"):
        n, ci, co, h, w = a[6:11]
        return f"conv3x3 fwd {ci}->{co} {h}x{w}", _conv(n, ci, co, h, w), None
    if fn in ("ppo_conv3x3_pool_forward_f32", "ppo_conv3x3_pool_forward_packed_f32"):
        n, ci, co, h, w = a[6:11]
        return f"conv3x3+maxpool fwd {ci}->{co} {h}x{w}", _conv(n, ci, co, h, w), None
    if fn in ("ppo_conv3x3_backward_data_f32", "ppo_conv3x3_backward_data_packed_f32"):
        n, ci, co, h, w = a[5:10]
        return f"conv3x3 bwd-data {co}->{ci} {h}x{w}", _conv(n, ci, co, h, w), None
    if fn == "ppo_conv3x3_backward_weight_slabs_f32":
        n, ci, co, h, w = a[5:10]
        return f"conv3x3 wgrad {ci}->{co} {h}x{w}", _conv(n, ci, co, h, w), None
    if fn == "ppo_conv3x3_backward_weight_slabs_pooled_f32":
        n, ci, co, h, w = a[6:11]
        return f"conv3x3 wgrad+pool-bwd {ci}->{co} {h}x{w}", _conv(n, ci, co, h, w), None
    if fn in ("ppo_conv3x3_backward_weight_slabs_batch_f32", "ppo_conv3x3_backward_weight_slabs_batch_mixed_f32"):
        k, n, ci, co, h, w = a[5:11]
        return f"conv3x3 wgrad x{k} {ci}->{co} {h}x{w}", k * _conv(n, ci, co, h, w), None
    if fn == "ppo_conv3x3_pool_bf16x3":
        n, ci, co, h, w = a[6:11]
        return f"conv3x3+maxpool fwd (3 x bf16 MFMA) {ci}->{co} {h}x{w}", _conv(n, ci, co, h, w), None
    if fn == "ppo_conv3x3_bf16x3":
        n, ci, co, h, w = a[5:10]
        kind = "fwd" if a[3] is not None else "bwd-data"
        return f"conv3x3 {kind} (3 x bf16 MFMA) {ci}->{co} {h}x{w}", _conv(n, ci, co, h, w), None
    if fn == "ppo_conv3x3_backward_weight_slabs_batch_bf16x3":
        k, n, ci, co, h, w = a[5:11]
        return f"conv3x3 wgrad x{k} (3 x bf16 MFMA) {ci}->{co} {h}x{w}", k * _conv(n, ci, co, h, w), None
    if fn == "ppo_conv3x3_backward_weight_f32":
        n, ci, co, h, w = a[7:12]
        return f"conv3x3 wgrad {ci}->{co} {h}x{w}", _conv(n, ci, co, h, w), None
    if fn == "ppo_impala_stack_tail_forward_f32":
        n, c, h, w = a[7:11]
        return f"stack blocks fwd (4 conv) {c}ch {h}x{w}", 4 * _conv(n, c, c, h, w), None
    if fn in ("ppo_impala_stack_tail_forward_bf16x3", "ppo_impala_stack_tail_backward_bf16x3",
              "ppo_impala_stack_tail_forward_signs_bf16x3", "ppo_impala_stack_tail_backward_signs_bf16x3"):
        n, c, h, w = a[8:12] if fn == "ppo_impala_stack_tail_forward_signs_bf16x3" else a[7:11]
        kind = "fwd" if "forward" in fn else "bwd-data"
        return f"stack blocks {kind} (4 conv, 3 x bf16 MFMA) {c}ch {h}x{w}", 4 * _conv(n, c, c, h, w), None
    if fn == "ppo_impala_stack_tail_backward_f32":
        n, c, h, w = a[7:11]
        return f"stack blocks bwd-data (4 conv) {c}ch {h}x{w}", 4 * _conv(n, c, c, h, w), None
    if fn == "ppo_impala_stack_full_forward_f32":
        n, c, h, w = a[9:13]
        return (f"whole stack fwd (5 conv + pool) {c}ch {h}x{w}",
                _conv(n, c, c, h, w) + 4 * _conv(n, c, c, _half(h), _half(w)), None)
    if fn == "ppo_impala_stack_chain_forward_f32":
        n, c, h, w = a[15:19]
        return (f"chained stacks fwd (9 conv + pool) {c}ch {h}x{w}",
                5 * _conv(n, c, c, h, w) + 4 * _conv(n, c, c, _half(h), _half(w)), None)
    if fn == "ppo_impala_stack_chain_split_forward_f32":  # (in, pre_w, pre_b, w, b, out, ws, ws_bytes, n, channels, h, w):
        n, c, h, w = a[8:12]                                # the same layers, two workgroups per image (inference)
        return (f"chained stacks fwd (9 conv + pool) {c}ch {h}x{w}",
                5 * _conv(n, c, c, h, w) + 4 * _conv(n, c, c, _half(h), _half(w)), None)
    if fn == "ppo_impala_stack_full_backward_f32":
        n, c, h, w = a[10:14]
        return (f"whole stack bwd-data (5 conv + pool) {c}ch {h}x{w}",
                _conv(n, c, c, h, w) + 4 * _conv(n, c, c, _half(h), _half(w)), None)
    if fn in ("ppo_impala_stack16_forward_f32", "ppo_impala_stack16_backward_f32"):
        n, c, h, w = a[-4:]
        kind = "fwd" if fn.endswith("forward_f32") else "bwd-data"
        return f"stack blocks {kind} (4 conv, sliding window) {c}ch {h}x{w}", 4 * _conv(n, c, c, h, w), None
    if fn == "ppo_gemm_f32":
        M, N, K = a[12:15]
        return f"gemm {M}x{N}x{K}", 2.0 * M * N * K, None
    if fn == "ppo_dense_heads_forward_f32":  # (x, relu_x, W, b, Wh, bh, relu_h, h, heads, M, K, H, NH, ws, ws_bytes)
        M, K, H, NH = a[9:13]
        return f"dense + heads fwd {M}x{K}x{H} (+{NH})", 2.0 * M * H * (K + NH), None
    if fn == "ppo_dense_heads_loss_forward_f32":  # the same + the PPO loss's arguments
        M, K, H, NH = a[9:13]
        return f"dense + heads + PPO loss fwd {M}x{K}x{H} (+{NH})", 2.0 * M * H * (K + NH), None
    if fn == "ppo_dense_heads_act_forward_f32":  # the same + (n_actions, temperature, seed, offset, outputs...)
        M, K, H, NH = a[9:13]
        return f"dense + heads + action sampling fwd {M}x{K}x{H} (+{NH})", 2.0 * M * H * (K + NH), None
    if fn == "ppo_conv3x3_block_forward_packed_f32":  # (in, pk0, b0, pk1, b1, out, n, c, h, w)
        n, c, h, w = a[6:10]
        return f"residual block fwd (2 conv) {c}ch {h}x{w}", 2 * _conv(n, c, c, h, w), None
    if fn in ("ppo_mlp_forward_f32", "ppo_mlp_train_f32"):  # (x, net*, [grads*,] index, [x_indexed,] B, ...)
        import ctypes
        from ppo_amd import _lib
        net = ctypes.cast(a[1], ctypes.POINTER(_lib.MlpNet)).contents
        F, H, NH = net.F, net.H, net.NH
        if fn == "ppo_mlp_forward_f32":
            B = a[3]
            return f"mlp forward (fc1, fc2, heads) {B}x{F}x{H} (+{NH})", 2.0 * B * (F * H + H * H + H * NH), None
        B = a[5]
        fwd = 2.0 * B * (F * H + H * H + H * NH)
        return (f"mlp forward + loss + backward, 2 launches {B}x{F}x{H} (+{NH})",
                fwd + 2.0 * B * (H * NH + H * H) + fwd, None)
    if fn == "ppo_adam_step_presummed_f32":
        return "clip + Adam (sums of g^2 from the gradient launch)", None, 28.0 * a[4]
    if fn == "ppo_gather_rows":
        return "gather observation rows", None, 2.0 * a[1] * a[4]
    if fn == "ppo_adam_step_f32":
        return "grad-norm + clip + Adam", None, 28.0 * a[4]
    if fn == "ppo_maxpool3x3s2_forward_f32":
        n, c, h, w = a[3:7]
        return f"maxpool fwd {c}ch {h}x{w}", None, n * c * (f32 * h * w + 5.0 * _half(h) * _half(w))
    if fn == "ppo_maxpool3x3s2_backward_f32":
        n, c, h, w = a[3:7]
        return f"maxpool bwd {c}ch {h}x{w}", None, n * c * (f32 * h * w + 5.0 * _half(h) * _half(w))
    if fn == "ppo_gae_scan_f32":
        return "gae scan", None, float(SCAN_BYTES_PER_ELEM) * a[7] * a[8]
    short = {"ppo_conv3x3_wgrad_reduce_f32": "wgrad slab reduction", "ppo_conv3x3_pack_weights_f32": "weight pack",
             "ppo_policy_act_f32": "policy sampling", "ppo_ppo_loss_f32": "PPO loss", "ppo_colsum_f32": "column sums",
             "ppo_moments_f64": "advantage moments", "ppo_normalize_f32": "advantage normalise",
             "ppo_heads_backward_f32": "heads backward (dh, dW, db, dense db)"}
    return short.get(fn, fn), None, None


class KernelTable:
    """Every C-ABI launch of one (untimed) iteration bracketed by HIP events on its launch stream, grouped by
    kernel class.  Durations of kernels that share the chip with another stream's kernels (backward-data next to
    the weight gradients; the two rollout groups) include that sharing, as in a rocprofv3 trace."""

    def __init__(self, objs, train_batch=256):
        self.rows = {}
        self.train_batch = train_batch
        self._undo = []
        for o in objs:
            orig = o._call
            o._call = self._wrap(orig)
            self._undo.append((o, orig))

    def _wrap(self, orig):
        import torch

        def call(fn_name, *a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig(fn_name, *a)
            e1.record()
            label, flops, nbytes = describe_call(fn_name, a, self.train_batch)
            row = self.rows.setdefault(label, {"events": [], "flops": 0.0, "bytes": 0.0})
            row["events"].append((e0, e1))
            row["flops"] += flops or 0.0
            row["bytes"] += nbytes or 0.0
        return call

    def remove(self):
        for o, orig in self._undo:
            o._call = orig

    def table(self, top=32):
        out, total = [], 0.0
        for label, row in self.rows.items():
            ms = sum(a.elapsed_time(b) for a, b in row["events"])
            total += ms
            out.append((ms, label, row))
        out.sort(reverse=True)
        rows = []
        for ms, label, row in out[:top]:
            r = {"kernel": label, "share": round(ms / total, 4), "launches": len(row["events"]),
                 "avg_us": round(1e3 * ms / len(row["events"]), 2)}
            if row["flops"]:
                r.update(bound="mfma", achieved=round(row["flops"] / (ms * 1e-3) / 1e12, 2), unit="TFLOP/s")
                peak = MFMA_BF16X3_PEAK_TFLOPS if "bf16" in label else MFMA_F32_PEAK_TFLOPS
                r["frac"] = round(r["achieved"] / peak, 4)
                if "bf16" in label:
                    r["peak"] = round(peak, 1)
            elif row["bytes"]:
                r.update(bound="hbm", achieved=round(row["bytes"] / (ms * 1e-3) / 1e9, 1), unit="GB/s")
                r["frac"] = round(r["achieved"] / HBM_PEAK_GBPS, 4)
            else:
                r.update(bound="latency")
            rows.append(r)
        mfma_ms = sum(ms for ms, _l, row in out if row["flops"])
        mfma_flops = sum(row["flops"] for _ms, _l, row in out)
        return {"kernel_ms_total": round(total, 2), "rows": rows,
                "mfma_kernels": {"share": round(mfma_ms / total, 4),
                                 "time_weighted_frac_of_peak": round(mfma_flops / (mfma_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)},
                "note": "one extra iteration of the same run; per-launch HIP events on the launch stream; kernels that "
                        "overlap another stream's kernels carry that sharing in their duration"}


# ----------------------------------------------------------------------------- GAE scan section
def bench_scan(lib, N, A_big, A_cfg):
    import numpy as np
    import torch
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


# ----------------------------------------------------------------------------- TVF returns section
def bench_tvf(K_heads=108, N=256, A=256, reps=20):
    """ppo_tvf_returns_f32 (rl/returns_truncated.py:623-693) at SURVEY.md §8(d)'s size: N = A = 256, K = V = 108
    geometric heads out to 30 000, 8 exponential n-step samples per head (`advanced` mode, n_step 20: the
    reference's TVF defaults).  Algorithmic bytes = 4 (N+1) A V read + 4 N A K written; HIP events on the launch
    stream around each launch; checked bit for bit against the NumPy oracle on a sample of env columns."""
    import numpy as np
    import torch
    from ppo_amd import returns_truncated as RT
    from ppo_amd.tvf import get_value_head_horizons
    dev = torch.device("cuda")
    hz = get_value_head_horizons(K_heads, 30000)
    K = V = len(hz)
    rng = np.random.default_rng(1)
    rewards = rng.normal(size=(N, A)).astype(np.float32)
    dones = rng.random((N, A)) < 0.01
    vs = rng.normal(size=(N + 1, A, V)).astype(np.float32)
    vs[:, :, 0] = 0
    np.random.seed(3)
    samples = RT._draw_samples("exponential", "advanced", N, np.asarray(hz), 20, 8, None)
    plan = RT.SampledReturnPlan(N, A, hz, hz, samples, False, dev)
    r, d, v = (torch.from_numpy(x).to(dev) for x in (rewards, dones.view(np.uint8), vs))
    out = torch.empty((N, A, K), dtype=torch.float32, device=dev)
    for _ in range(3):
        plan.launch(0.999, r, d, v, out)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for k in range(reps):
        plan.launch(0.999, r, d, v, out)
        ev[k + 1].record()
    torch.cuda.synchronize()
    per = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(reps))
    ms = sum(per) / reps
    nbytes = 4 * (N + 1) * A * V + 4 * N * A * K
    from oracle import returns_truncated as T
    cols = np.arange(0, A, max(1, A // 8))[:8]
    ref = T.sampled_returns(0.999, rewards[:, cols], dones[:, cols], hz, hz, vs[:, cols], samples)
    gbps = nbytes / (ms * 1e-3) / 1e9
    return {"N": N, "A": A, "K": K, "V": V, "samples_per_head": int(samples.shape[1]), "distinct_n": plan.ND,
            "max_n": plan.max_n, "algorithmic_bytes": nbytes, "avg_kernel_us": round(ms * 1e3, 2),
            "min_kernel_us": round(per[0] * 1e3, 2), "max_kernel_us": round(per[-1] * 1e3, 2), "launches_timed": reps,
            "achieved_GBps": round(gbps, 1), "peak_GBps": HBM_PEAK_GBPS, "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBPS, 4),
            "terms": N * A * K * int(samples.shape[1]),
            "kernel": "tvf_column_kernel (one workgroup per env column, value samples resident in LDS)"
                      if os.environ.get("PPO_AMD_TVF_COLUMN", "1") != "0" else "tvf_prefix_kernel + tvf_gather_kernel",
            "bit_exact_vs_oracle": bool(np.array_equal(out[:, cols].cpu().numpy(), ref)),
            "bound": "VALU issue (about 15 wave instructions per (t, a, k, c) term: the float64 blend of NumPy's "
                     "promotion rules), not HBM; see DESIGN.md §4"}


# ----------------------------------------------------------------------------- CPU baseline
def cpu_baseline(N, A, epochs, mb, obs_shape=(4, 84, 84), n_actions=6):
    """The PPO iteration on the host through plain torch CPU ops (oracle/model_torch.py), bounded: one rollout
    forward of `A` observations and one train minibatch of `mb` samples (the GPU workload's own batch sizes) are
    timed, then extrapolated to a full iteration of N*A env steps."""
    import numpy as np
    import torch
    from oracle import model_torch as R, returns as O
    from ppo_amd.models import ImpalaSpec, init_impala_parameters
    threads = torch.get_num_threads()
    torch.manual_seed(1)
    init = init_impala_parameters(ImpalaSpec(tuple(obs_shape)), n_actions, 1, 0.1, True)
    fb, tb = A, mb
    rng = np.random.default_rng(0)
    xf = torch.from_numpy(rng.integers(0, 256, (fb, *obs_shape), dtype=np.uint8))
    xt = torch.from_numpy(rng.integers(0, 256, (tb, *obs_shape), dtype=np.uint8))
    actions = torch.from_numpy(rng.integers(0, n_actions, (tb,)))
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
        while reps < 1 or (time.perf_counter() - t0 < 6.0 and reps < 20):
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


def cpu_baseline_mlp(N, A, mb, F, H, n_actions, K, epochs):
    """configs[4] on the host through plain torch CPU ops (oracle/model_torch.mlp_forward): one rollout forward of both
    nets over A observations, one minibatch step of each phase's network (forward, a squared-error loss over the heads
    that phase trains, backward, clip, Adam) and the NumPy TVF-return oracle on a slice of the envs, extrapolated to one
    iteration (`epochs` = (policy, value, distil) epochs)."""
    import numpy as np
    import torch
    from oracle import model_torch as R, returns as O, returns_truncated as OT
    from ppo_amd.models import MLPSpec, init_parameters
    from ppo_amd import tvf as tvf_mod
    threads = torch.get_num_threads()
    torch.manual_seed(1)
    spec = MLPSpec((F,), hidden_units=H)
    nets = [init_parameters(spec, n_actions, 1, 0.1, True, K) for _ in range(2)]
    rng = np.random.default_rng(0)
    xf = torch.from_numpy(rng.normal(size=(A, F)).astype(np.float32))
    xt = torch.from_numpy(rng.normal(size=(mb, F)).astype(np.float32))
    tgt = torch.from_numpy(rng.normal(size=(mb, K)).astype(np.float32))
    sds = [{k: v.clone().requires_grad_(True) for k, v in n.items()} for n in nets]
    opts = [torch.optim.Adam(list(sd.values()), lr=2.5e-4, eps=1e-5) for sd in sds]

    def fwd():
        with torch.no_grad():
            for n in nets:
                R.mlp_forward(n, xf)

    def step(i):
        def run():
            opts[i].zero_grad(set_to_none=True)
            out = R.mlp_forward(sds[i], xt)
            loss = 0.5 * torch.square(out["tvf_value"] - tgt).mean() + torch.square(out["raw_policy"]).mean()
            loss.backward()
            torch.nn.utils.clip_grad_norm_([p for p in sds[i].values() if p.grad is not None], 20.0)
            opts[i].step()
        return run
    times = {}
    for name, fn in (("fwd", fwd), ("policy_net_step", step(0)), ("value_net_step", step(1))):
        fn()
        t0, reps = time.perf_counter(), 0
        while reps < 1 or (time.perf_counter() - t0 < 3.0 and reps < 200):
            fn()
            reps += 1
        times[name] = (time.perf_counter() - t0) / reps
    # returns: GAE (C oracle) + the TVF return estimator (NumPy oracle) on 16 env columns, scaled to A
    r = rng.normal(size=(N, A)).astype(np.float32)
    v = rng.normal(size=(N + 1, A)).astype(np.float32)
    d = rng.random((N, A)) < 0.01
    t0 = time.perf_counter()
    O.gae_and_returns(r, v[:N], v[N], d, 0.999, 0.95, 0.95)
    t_scan = time.perf_counter() - t0
    hz = np.asarray(tvf_mod.get_value_head_horizons(K, 30000, "geometric"))
    As = 16
    vs = rng.normal(size=(N + 1, As, len(hz))).astype(np.float32)
    samples = rng.integers(1, 41, size=(len(hz), 8))
    t0 = time.perf_counter()
    OT.sampled_returns(0.999, r[:, :As], d[:, :As], hz, hz, vs, samples)
    t_tvf = (time.perf_counter() - t0) * A / As
    n_mb = N * A // mb
    per_iter = ((N + 1) * times["fwd"] + (epochs[0] + epochs[2]) * n_mb * times["policy_net_step"]
                + epochs[1] * n_mb * times["value_net_step"] + t_scan + t_tvf)
    return {"value": round(N * A / per_iter, 1), "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"torch-CPU restatement (oracle/model_torch.mlp_forward): 1 rollout forward of both nets over {A} obs "
                      f"({times['fwd']*1e3:.2f} ms), 1 minibatch step of {mb} per net ({times['policy_net_step']*1e3:.2f} / "
                      f"{times['value_net_step']*1e3:.2f} ms), C-oracle GAE scan ({t_scan*1e3:.1f} ms), NumPy TVF-return oracle on "
                      f"{As} of {A} env columns ({t_tvf*1e3:.0f} ms scaled), extrapolated to one iteration of {N*A} env steps "
                      f"({epochs[0]} policy / {epochs[1]} value / {epochs[2]} distil epochs)",
            "host_cores_available": os.cpu_count()}


# ----------------------------------------------------------------------------- main
def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a)
    # first thing in a rank, before any HIP call / thread pool / pinned allocation: the cores of this GPU's NUMA node
    from ppo_amd import affinity
    pinned_cpus = affinity.pin_rank(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                                    enabled=not a.no_affinity)
    world, rank, local = init_dist(a)
    import numpy as np
    import torch
    if a.rendezvous_only:
        t = torch.ones(1)
        if world > 1:
            torch.distributed.all_reduce(t)
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "n_gpus": a.gpus, "ranks_seen": int(t.item()),
                              "backend": torch.distributed.get_backend() if world > 1 else "none"}), flush=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        return 0
    from ppo_amd import _lib, envs, logger, models, parallel, rollout
    from ppo_amd.config import args
    lib = _lib.load()
    cfg = CONFIGS[a.config]
    N, A = a.n_steps, a.agents or cfg["agents"]
    if a.scan_only:
        print(json.dumps({"gae_scan": bench_scan(lib, N, a.scan_envs, A)}), flush=True)
        return 0
    if a.tvf_only:
        print(json.dumps({"tvf_returns": bench_tvf(a.tvf_heads)}), flush=True)
        return 0
    mb = 256
    args.setup([f"--agents={A}", f"--n_steps={N}", *cfg["flags"], "--seed=1", f"--device=cuda:{local}",
                f"--policy_opt_mini_batch_size={mb * world}", "--disable_logging=True", "--upload_batch=True",
                "--env_reward_normalization=off", f"--precision={a.precision}"])
    torch.manual_seed(1)
    np.random.seed(1 + rank)
    if a.config == "humanoid_tvf":
        # the float / gaussian synthetic env in the Humanoid's shape, as two array-stepping groups (pipelined rollout)
        from ppo_amd import tvf as tvf_mod
        from ppo_amd.vec_env import SplitVecEnv, SyntheticFloatVecEnv
        obs_shape, n_actions = cfg["obs"], cfg["actions"]
        horizons, weights = tvf_mod.get_value_head_horizons(args.tvf.value_heads, args.tvf.max_horizon,
                                                            args.tvf.head_spacing, include_weight=True)
        args.tvf.value_heads = len(horizons)
        model = models.TVFModel(encoder="mlp", input_dims=obs_shape, actions=n_actions, device=f"cuda:{local}",
                                architecture="dual", hidden_units=args.model.hidden_units, encoder_activation_fn="tanh",
                                tvf_fixed_head_horizons=horizons, tvf_fixed_head_weights=weights,
                                head_scale=args.model.head_scale, head_bias=args.model.head_bias)
        runner = rollout.Runner(model, logger.Logger(quiet=True), action_dist="gaussian")
        runner.vec_env = SplitVecEnv([SyntheticFloatVecEnv(A // 2, obs_shape[0], n_actions, seed=1,
                                                           env_offset=rank * A + i * (A // 2)) for i in range(2)])
    else:
        obs_shape, n_actions = envs.get_env_spec()
        assert tuple(obs_shape) == tuple(cfg["obs"]) and n_actions == cfg["actions"], (obs_shape, n_actions)
        model = models.TVFModel(encoder="impala", input_dims=obs_shape, actions=n_actions, device=f"cuda:{local}",
                                architecture="single", hidden_units=args.model.hidden_units,
                                head_scale=args.model.head_scale, head_bias=args.model.head_bias, precision=args.precision)
        runner = rollout.Runner(model, logger.Logger(quiet=True))
        runner.vec_env = envs.create_envs_classic(rank=rank, world=world)
    runner.reset()
    if os.environ.get("PPO_AMD_DUMP_MAPS"):
        # diagnostics for crashes under a profiler: the load addresses that turn a raw stack trace into library + offset
        with open(os.environ["PPO_AMD_DUMP_MAPS"], "w") as f:
            f.write(open("/proc/self/maps").read())

    # The roofline probe (pong): the kernel with the largest share of GPU time in profiles/ — the training forward's
    # chained LDS-resident launch (residual blocks of the 21x21 stack + the whole 11x11 stack); with the fused paths
    # switched off (PPO_AMD_FUSE_STACK_TAIL / _CHAIN = 0) it falls back to the stack-tail launch or the single 32->32
    # 21x21 forward convolution.  Minibatch-sized training launches (n == mb) and the rollout groups' half-batch launches
    # (n == A / 2) are timed separately: at half batch the one-workgroup-per-image kernels fill half the chip.
    # The other configs take their roofline kernel from the per-kernel table below (its largest row).
    geo = {"pong": (32, 21, 21)}.get(a.config)
    split = bool(getattr(model.policy_net, "split_bf16", False))
    if split:
        geo = None  # the probed launches do not run in split mode: the per-kernel table's largest row is the roofline kernel

    def probe_match_n(n_want):
        def match(fn_name, c):
            if geo is None:
                return False
            if fn_name == "ppo_impala_stack_chain_forward_f32":  # (in, pre_w, pre_b, pre_a0..pre_q1, w, b, pooled, argmax,
                return (c[15], c[16], c[17], c[18]) == (n_want, *geo)                 # a0..q1, n, channels, h, w)
            if fn_name == "ppo_impala_stack_tail_forward_f32":  # (in, w[4], b[4], a0, q0, a1, q1, n, channels, h, w)
                return (c[7], c[8], c[9], c[10]) == (n_want, *geo)
            return c[1] == 1 and (c[6], c[7], c[8], c[9], c[10]) == (n_want, 32, *geo)
        return match

    probe_fns = ("ppo_conv3x3_forward_f32", "ppo_conv3x3_forward_packed_f32", "ppo_impala_stack_tail_forward_f32",
                 "ppo_impala_stack_chain_forward_f32")
    probe = CallProbe(model.policy_net, probe_fns, probe_match_n(mb))
    conv_flops = 2 * 9 * 32 * 32 * 21 * 21 * mb

    def iteration():
        runner.generate_rollout()
        runner.calculate_returns()
        runner.train()

    for _ in range(a.warmup):
        iteration()
    reducer = runner._reducers.get(id(runner.net))
    if reducer is not None:
        reducer.exposed_ms()  # drop the warm-up's marks
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
    probe.remove()
    stats = runner.fetch_stats()
    exposed_comm_ms = reducer.exposed_ms() if reducer is not None else None

    env_steps = world * N * A * a.steps
    roofline = None
    if geo is not None:
        if not probe.events:
            raise SystemExit("bench.py: the roofline probe saw no launch of its kernel (entry point renamed?)")
        kern_ms = probe.avg_ms()
        map_bytes = 32 * 21 * 21 * 4 * mb
        small_bytes = 32 * 11 * 11 * 4 * mb
        if probe.seen == "ppo_impala_stack_chain_forward_f32":
            # the 21x21 stack's 4 block convolutions, the 11x11 stack's first convolution (on the 21x21 map) + max-pool
            # and its 4 block convolutions; reads the 21x21 pooled map, writes the 4 + 4 maps the backward pass needs,
            # the pooled 11x11 map and its uint8 argmax
            probe_flops = 5 * conv_flops + 4 * (2 * 9 * 32 * 32 * 11 * 11 * mb)
            probe_bytes = 5 * map_bytes + 5 * small_bytes + small_bytes // 4
            probe_kernel = ("stack_full_kernel<32,21,21 -> 11,11> chained training forward: residual blocks of the 21x21 "
                            "stack + the whole 11x11 stack (9 convolutions + max-pool) in one launch, maps resident in LDS "
                            "(ppo_impala_stack_chain_forward_f32, minibatch launches)")
            traffic_key = "stack_full_kernel"
        elif probe.seen == "ppo_impala_stack_tail_forward_f32":   # reads the block input, writes a0, q0, a1, q1
            probe_flops, probe_bytes = 4 * conv_flops, 5 * map_bytes
            probe_kernel = ("stack_tail_kernel<32,21,21> training forward: the 4 residual-block convolutions of the 21x21 "
                            "stack in one launch (ppo_impala_stack_tail_forward_f32, minibatch launches)")
            traffic_key = "stack_tail_kernel"
        else:
            probe_flops, probe_bytes = conv_flops, 2 * map_bytes
            probe_kernel = "conv3x3_kernel<32,32,21,21,IN_RELU> forward (ppo_conv3x3_forward_f32, minibatch launches)"
            traffic_key = "conv3x3_kernel"
        traffic, traffic_source = None, "no PMC pass over this command on file (tools/pmc_bench.sh)"
        if mb == 256 and os.path.exists(TRAFFIC_FILE):
            doc = json.load(open(TRAFFIC_FILE))
            rec = doc.get(traffic_key)
            if rec:
                # the counters were taken over a particular build of the kernels: say which, and report null when the
                # kernel sources have changed since (profiles/*_bench_hbm_traffic.json carries their hash)
                now = kernel_source_hash()
                if doc.get("kernel_source_sha16") in (None, now):
                    traffic = int(rec["hbm_bytes_per_launch"])
                    traffic_source = os.path.relpath(TRAFFIC_FILE, ROOT) + ": " + rec.get("how", "") + (
                        "" if doc.get("kernel_source_sha16") else " [no kernel-source hash on file: an earlier round's pass]")
                else:
                    traffic_source = (f"{os.path.relpath(TRAFFIC_FILE, ROOT)} was measured on kernel sources "
                                      f"{doc.get('kernel_source_sha16')} (now {now}): stale, not reported")
        tflops = probe_flops / (kern_ms * 1e-3) / 1e12
        roofline = {"bound": "mfma", "achieved": round(tflops, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tflops / MFMA_F32_PEAK_TFLOPS, 4),
                    # HBM bytes per launch of this kernel from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate
                    # runs) over this same bench.py command, 2 x FETCH_SIZE + WRITE_SIZE KiB as the guide prescribes
                    "traffic": traffic, "traffic_source": traffic_source,
                    "algorithmic_bytes_per_launch": probe_bytes, "kernel": probe_kernel,
                    "algorithmic_flops_per_launch": probe_flops, "avg_kernel_ms": round(kern_ms, 4),
                    "launches_timed": len(probe.events)}
    fwd_mflop = forward_mflop(obs_shape, n_actions, args.model.hidden_units, a.config)
    epochs_fwd = args.policy_opt.epochs + (args.value_opt.epochs + args.distil_opt.epochs if runner.dual else 0)
    samples_fwd = (N + 1) * A * (2 if runner.dual else 1) + epochs_fwd * N * A
    samples_bwd = epochs_fwd * N * A
    model_tflops = (samples_fwd + 2 * samples_bwd) * fwd_mflop * 1e6 * a.steps / wall / 1e12
    obs_kind = "uint8" if len(obs_shape) == 3 else "float32"
    net_kind = ("IMPALA-CNN single architecture" if a.config != "humanoid_tvf" else
                f"tanh MLP dual architecture (policy + value + distil phases), {args.tvf.value_heads} TVF heads, gaussian policy")
    out = {
        "metric": cfg["metric"],
        "value": round(env_steps / wall, 1),
        "unit": "env-steps/s",
        "n_gpus": world,
        "ranks_seen": parallel.world_size(),
        "cpu_affinity": ({"cpus": len(pinned_cpus), "first": pinned_cpus[0], "last": pinned_cpus[-1]} if pinned_cpus
                         else "unchanged (--no-affinity, one NUMA node, or no topology in sysfs)"),
        "backend": parallel.backend_name(),
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(wall / a.steps * 1e3, 2),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if not split else "f32 + bf16x3 (opt-in --precision=%s: every 16- / 32-channel convolution - residual blocks, "
                                         "stack-first convolutions + max-pool; forward, backward-data, weight gradients - as 3 bf16 "
                                         "MFMAs per product with f32 accumulation; the uint8 first layer, dense layer, heads, losses, "
                                         "Adam f32)" % a.precision,
        "data": "synthetic",
        "config": {"workload": f"{a.config}: PPO iteration, {A} envs/GPU x {N} steps, obs {tuple(obs_shape)} {obs_kind}, "
                               f"{n_actions} actions, {net_kind} ({model.model_size()} params), "
                               f"{args.policy_opt.epochs} policy epochs, global minibatch {mb * world}, Adam, synthetic env, "
                               f"reward normalisation off (the synthetic env's rewards are N(0,1) already; the reference's "
                               f"default 'rms' is a host-side vector wrapper, rl/config.py:508)",
                   "reward_normalization": "off", "envs_per_gpu": A, "n_steps": N, "global_minibatch": mb * world, "parallelism": f"dp{world}"},
        "roofline": roofline,
        "phase_seconds_per_step": {k: round(v / a.steps, 4) for k, v in phase.items()},
        "model_tflops_whole_step": round(model_tflops / world, 2),
        "model_frac_of_mfma_peak_whole_step": round(model_tflops / world / MFMA_F32_PEAK_TFLOPS, 4),
        "train_stats": {k: round(float(v), 6) for k, v in stats.items()},
    }
    if world > 1:
        out["grad_allreduce"] = {"buckets": 2, "exposed_ms_per_optimizer_step": None if exposed_comm_ms is None
                                 else round(exposed_comm_ms, 4),
                                 "note": "time the compute stream waited between 'backward queued' and 'gradients reduced'"}
    if not a.no_kernel_table:
        # the time-weighted picture: one more iteration of the same run with every launch bracketed (untimed; the
        # rollout's recorded launch plans are dropped so that its launches pass through the bracket too)
        nets = list({id(n): n for n in (model.policy_net, model.value_net)}.values())
        for n_ in nets:
            n_.use_plans = False
        kt = KernelTable([runner] + nets, train_batch=mb)
        iteration()
        torch.cuda.synchronize()
        kt.remove()
        for n_ in nets:
            n_.use_plans = True
        table = kt.table()
        out["roofline_by_kernel"] = table
        if roofline is not None:
            # the same layers at the ROLLOUT's geometry (half-batch env groups: 128 images for 256 CUs; run as the
            # two-workgroups-per-image launch unless PPO_AMD_CHAIN_SPLIT=0), so that the quoted fraction is not the
            # flattering half of their launches
            for row in table["rows"]:
                if row["kernel"].startswith("chained stacks fwd") and "[n=" in row["kernel"] and "frac" in row:
                    roofline["rollout_geometry"] = {k_: row[k_] for k_ in ("kernel", "launches", "avg_us", "achieved", "frac")}
                    tr = [r_ for r_ in table["rows"] if r_["kernel"] == row["kernel"].split(" [n=")[0]]
                    if tr:
                        n_all = row["launches"] + tr[0]["launches"]
                        roofline["blended_frac_all_launches"] = round(
                            (row["frac"] * row["launches"] * row["avg_us"] + tr[0]["frac"] * tr[0]["launches"] * tr[0]["avg_us"])
                            / (row["launches"] * row["avg_us"] + tr[0]["launches"] * tr[0]["avg_us"]), 4)
        else:
            # configs without a hand-picked probe: the table's largest roofline-bound row is the roofline kernel
            top = next((r_ for r_ in table["rows"] if "frac" in r_), None)
            if top is not None:
                out["roofline"] = {"bound": top["bound"], "achieved": top["achieved"],
                                   "peak": top.get("peak", MFMA_F32_PEAK_TFLOPS) if top["bound"] == "mfma" else HBM_PEAK_GBPS,
                                   "unit": top["unit"], "frac": top["frac"], "traffic": None,
                                   "kernel": top["kernel"] + " (largest roofline-bound share of kernel time, "
                                             f"{top['share']:.1%}; per-launch HIP events of the extra iteration)",
                                   "avg_kernel_ms": round(top["avg_us"] * 1e-3, 4), "launches_timed": top["launches"]}
    if world == 1 and a.precision == "high" and a.config != "humanoid_tvf" and not a.no_opt_in:
        # The reference's own default is --precision=medium (TF32 convolutions on its GPUs, /root/reference train.py:166-178
        # as SURVEY.md reads it); here that flag selects the split-bf16 launches (~16-bit products).  Timed on the same
        # workload in the same process right after the exact-float32 measurement, reported BESIDE the line: `value`
        # above is exact float32 and stays so.
        try:
            args.precision = "medium"
            model2 = models.TVFModel(encoder="impala", input_dims=obs_shape, actions=n_actions, device=f"cuda:{local}",
                                     architecture="single", hidden_units=args.model.hidden_units,
                                     head_scale=args.model.head_scale, head_bias=args.model.head_bias, precision="medium")
            runner2 = rollout.Runner(model2, logger.Logger(quiet=True))
            runner2.vec_env = envs.create_envs_classic(rank=rank, world=world)
            runner2.reset()
            for _ in range(max(1, a.warmup)):
                runner2.generate_rollout(), runner2.calculate_returns(), runner2.train()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                runner2.generate_rollout(), runner2.calculate_returns(), runner2.train()
            torch.cuda.synchronize()
            wall2 = time.perf_counter() - t0
            out["opt_in_precision"] = {
                "flag": "--precision=medium (the reference's default flag value; bench.py --precision medium gives the full line)",
                "value": round(N * A * a.steps / wall2, 1), "unit": "env-steps/s", "ms_per_step": round(wall2 / a.steps * 1e3, 2),
                "steps": a.steps, "split_launches": bool(getattr(model2.policy_net, "split_bf16", False)),
                "dtype": "f32 + bf16x3: every 16- / 32-channel convolution (residual blocks, stack-first convolutions + max-pool; forward, "
                         "backward-data, weight gradients) as 3 bf16 MFMAs per product with f32 accumulation (~16-bit products); the uint8 "
                         "first layer, dense layer, heads, losses, Adam f32",
                "vs_value": round(N * A * a.steps / wall2 / (env_steps / wall), 3)}
        except Exception as e:  # the opt-in leg must never cost the line its exact-float32 value
            out["opt_in_precision"] = {"flag": "--precision=medium", "error": f"{type(e).__name__}: {e}"[:400]}
        args.precision = "high"
    if rank == 0:
        if world == 1 and not a.no_scan:
            out["gae_scan"] = bench_scan(lib, N, a.scan_envs, A)
            out["tvf_returns"] = bench_tvf(a.tvf_heads)
        if world == 1 and not a.no_cpu_baseline:
            affinity.release()  # the CPU leg gets the whole host, not this rank's NUMA share (13 x slower when pinned)
            out["cpu_baseline"] = (cpu_baseline(N, A, args.policy_opt.epochs, mb, tuple(obs_shape), n_actions)
                                   if a.config != "humanoid_tvf" else
                                   cpu_baseline_mlp(N, A, mb, obs_shape[0], args.model.hidden_units, n_actions, runner.K,
                                                    (args.policy_opt.epochs, args.value_opt.epochs, args.distil_opt.epochs)))
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
