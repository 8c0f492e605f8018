"""CPU placement of one rank (one process per GPU) on a multi-socket node.

No counterpart in the reference (one job per GPU, placement left to slurm: runner.py).  Each rank of this trainer runs
16 env threads plus the pinned-memory uploads of the rollout on the critical path (~0.2 ms of host work per env-group
step), so with 8 ranks on a 2-socket node a rank whose threads float across sockets pays remote-memory
latency on every observation it writes and on the H2D copies out of its pinned blocks.  `pin_rank()` is called FIRST
thing in a rank - before any HIP call, before the env thread pool exists and before pinned buffers are allocated, so
that threads inherit the mask and pages are first-touched on the right node:

  * the GPU's NUMA node comes from sysfs (KFD topology -> drm render minor -> PCI device's `numa_node`);
  * the rank takes the CPUs of that node that the process may use, divided evenly among the local ranks whose GPUs sit
    on the same node (GPU order), so sibling ranks do not contend for cores either;
  * no NUMA information (single-node VMs report -1): the allowed CPUs are divided evenly among the local ranks;
  * `os.sched_setaffinity` in-process - never a re-exec (a process that touched the GPU must not exec).

Everything that reads sysfs takes a `root` argument so the parser is testable on a fake tree (tests/test_host_cpu.py).
"""
import os
from typing import Dict, List, Optional, Sequence


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    cpus: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-")
            cpus.extend(range(int(lo), int(hi) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def _props(path: str) -> Dict[str, str]:
    out = {}
    with open(path) as f:
        for ln in f:
            kv = ln.split()
            if len(kv) >= 2:
                out[kv[0]] = kv[1]
    return out


def gpu_numa_nodes(root: str = "/") -> List[int]:
    """NUMA node of every GPU in KFD order (the order HIP enumerates them in, before *_VISIBLE_DEVICES); -1 = unknown."""
    nodes_dir = os.path.join(root, "sys/class/kfd/kfd/topology/nodes")
    if not os.path.isdir(nodes_dir):
        return []
    gpus = []
    for d in sorted(os.listdir(nodes_dir), key=lambda s: int(s) if s.isdigit() else 1 << 30):
        try:
            p = _props(os.path.join(nodes_dir, d, "properties"))
        except OSError:
            continue
        if int(p.get("simd_count", "0")) <= 0:
            continue  # a CPU node
        numa = -1
        minor = p.get("drm_render_minor")
        if minor is not None:
            try:
                with open(os.path.join(root, f"sys/class/drm/renderD{minor}/device/numa_node")) as f:
                    numa = int(f.read().strip())
            except (OSError, ValueError):
                numa = -1
        gpus.append(numa)
    return gpus


def visible_gpu_indices(n_gpus: int, env=os.environ) -> List[int]:
    """Which KFD GPUs the local ranks 0.. map to, honouring ROCR_ / HIP_ / CUDA_VISIBLE_DEVICES index lists."""
    idx = list(range(n_gpus))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(var)
        if v is None:
            continue
        try:
            pick = [int(x) for x in v.split(",") if x.strip() != ""]
        except ValueError:  # UUIDs: order unknown, keep the plain order
            continue
        idx = [idx[i] for i in pick if 0 <= i < len(idx)]
    return idx


def node_cpus(node: int, root: str = "/") -> List[int]:
    try:
        with open(os.path.join(root, f"sys/devices/system/node/node{node}/cpulist")) as f:
            return parse_cpulist(f.read())
    except OSError:
        return []


def plan(local_rank: int, local_world: int, allowed: Sequence[int], root: str = "/", env=os.environ) -> Optional[List[int]]:
    """The CPUs rank `local_rank` of `local_world` should run on, or None when there is nothing to decide."""
    allowed = sorted(allowed)
    if local_world < 1 or not (0 <= local_rank < local_world) or len(allowed) < 2:
        return None
    numa = gpu_numa_nodes(root)
    vis = visible_gpu_indices(len(numa), env)
    mine = numa[vis[local_rank]] if local_rank < len(vis) else -1
    if mine >= 0:
        pool = [c for c in node_cpus(mine, root) if c in set(allowed)]
        # local ranks whose GPU sits on the same node share that node's cores evenly, in GPU order
        sharers = [r for r in range(min(local_world, len(vis))) if numa[vis[r]] == mine]
    else:
        pool, sharers = [], []
    if not pool:
        pool, sharers = allowed, list(range(local_world))
    k, n = sharers.index(local_rank) if local_rank in sharers else 0, max(len(sharers), 1)
    per = len(pool) // n
    if per < 1:
        return pool
    return pool[k * per:(k + 1) * per]


def pin_rank(local_rank: int, local_world: int, enabled: bool = True) -> Optional[List[int]]:
    """Pin this process (and every thread it starts from now on) as `plan` says.  Returns the CPU list applied, or None
    (disabled, a single rank with one NUMA node's worth of CPUs anyway, or an OS without sched_setaffinity)."""
    if not enabled or os.environ.get("PPO_AMD_AFFINITY", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        allowed = os.sched_getaffinity(0)
        cpus = plan(local_rank, local_world, allowed)
        if not cpus or set(cpus) == set(allowed):
            return None
        os.sched_setaffinity(0, cpus)
        global _unpinned_mask
        _unpinned_mask = set(allowed)
        return cpus
    except OSError:
        return None


_unpinned_mask = None  # the mask pin_rank found, for release()


def release() -> bool:
    """Hand every thread of this process the CPU mask it had before pin_rank (threads started while pinned inherited the
    narrow mask).  For host-only legs that are not part of a rank's hot path and should see the whole machine - bench.py's
    CPU baseline is timed this way, so that it is the host at its best and not a rank's NUMA share."""
    if _unpinned_mask is None or not hasattr(os, "sched_setaffinity"):
        return False
    ok = True
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = [0]
    for tid in tids:
        try:
            os.sched_setaffinity(tid, _unpinned_mask)
        except OSError:
            ok = False  # (a thread that exited meanwhile)
    return ok
