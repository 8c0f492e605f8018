"""Kernel timeline of the rollout out of a rocprofv3 --kernel-trace CSV: per env step, which kernels ran when, on which
queue, how much of the step the GPU was busy and how much of that with two kernels at once.
Usage: python tools/rollout_timeline.py <dir with *kernel_trace.csv> [steps to print]"""
import collections
import csv
import glob
import os
import re
import sys

src = sys.argv[1]
show = int(sys.argv[2]) if len(sys.argv) > 2 else 3
files = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\s+", "", r["Kernel_Name"].replace("ppo::(anonymous namespace)::", "").replace("void ", ""))
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0][:60], r.get("Queue_Id", "?"),
                     int(r.get("Grid_Size", 0) or 0) // max(int(r.get("Workgroup_Size", 1) or 1), 1)))
rows.sort()
print(len(rows), "dispatches; columns:", list(csv.DictReader(open(files[0])).fieldnames))
# an env step of one group ends with its sampling kernel
step_end = "policy_act" if any("policy_act" in r[2] for r in rows) else "finalize_heads"  # sampling fused into the heads launch
acts = [i for i, r in enumerate(rows) if step_end in r[2]]
print(step_end, "launches:", len(acts))
t_first = rows[acts[len(acts) // 2]][0]
sel = [r for r in rows if r[0] >= t_first][: 60 * show]
base = sel[0][0]
for s, e, n, q, wg in sel[: 40 * show]:
    print(f"{(s - base) / 1e3:9.1f} {(e - base) / 1e3:9.1f} {(e - s) / 1e3:7.1f} us  q{q:>3} wg{wg:5d}  {n}")
# busy / overlap over the second half of the trace
half = rows[len(rows) // 2:]
ev = []
for s, e, *_ in half:
    ev += [(s, 1), (e, -1)]
ev.sort()
busy = over = 0
depth, last = 0, ev[0][0]
for t, d in ev:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        over += t - last
    depth += d
    last = t
span = half[-1][1] - half[0][0]
n_steps = sum(1 for r in half if step_end in r[2]) / 2
print(f"second half: span {span / 1e6:.2f} ms, busy {busy / span:.1%}, two or more kernels at once {over / span:.1%}, "
      f"{span / 1e3 / max(n_steps, 1):.1f} us per env step over {n_steps:.0f} steps")
tot = collections.Counter()
for s, e, n, q, wg in half:
    tot[(n, wg)] += e - s
for (n, wg), t in tot.most_common(14):
    print(f"  {t / 1e3 / max(n_steps, 1):8.1f} us per env step  wg{wg:5d}  {n}")
