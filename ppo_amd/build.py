"""Build libppo_amd.so (HIP, gfx950 only) in-tree with hipcc.

    python -m ppo_amd.build [--force]

hipcc cross-compiles without a GPU.  One object per .hip file (rebuilt only
when the source or a header is newer), linked into ppo_amd/lib/libppo_amd.so.
The .so is git-ignored but travels with the tree to the GPU box.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(HERE, "lib", "libppo_amd.so")
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libppo_amd.so cannot be built")


def _newest_header() -> float:
    t = 0.0
    for d in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(d):
            if f.endswith((".h", ".hpp", ".cuh")):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    flags = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
             "-I", os.path.join(HERE, "..", "include")]
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdr_t = _newest_header()
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append([cc, *flags, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
