"""Generate tests/golden/wrappers_golden.npz by driving the REFERENCE's vector-env wrappers
(rl/wrappers.py VecNormalizeRewardWrapper :795-919, VecRepeatedActionPenalty :758-793) and
RunningMeanStd (rl/utils.py:416-455) over a scripted fake vector env.  Runs in the build container only
(needs /root/reference); the fixture it writes is data: inputs and expected outputs.

    python tests/golden/make_wrappers_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_shim import load_reference  # noqa: E402

rl = load_reference(["--env_type=atari", "--output_folder=/tmp"])
from rl import utils as ref_utils  # noqa: E402
from rl import wrappers as ref_wrappers  # noqa: E402

A, T = 6, 60
rng = np.random.default_rng(1234)
rewards = (rng.standard_normal((T, A)) * rng.choice([0.1, 1.0, 30.0], size=(T, A))).astype(np.float32)
dones = rng.random((T, A)) < 0.1
actions = rng.integers(-1, 3, size=(T, A)).astype(np.int32)
actions[10:24, 0] = 2  # a long run of one action in env 0
actions[30:40, 3] = 1


class ScriptedVecEnv:
    num_envs = A

    def __init__(self):
        self.t = 0

    def reset(self):
        self.t = 0
        return np.zeros((A, 1), np.float32)

    def step(self, a):
        t = self.t
        self.t += 1
        return np.zeros((A, 1), np.float32), rewards[t].copy(), dones[t].copy(), [{"time": t} for _ in range(A)]


out = {"rewards": rewards, "dones": dones, "actions": actions}
for tag, kw in (("rms", dict(gamma=0.999, clip=10.0)), ("rms_scale", dict(gamma=0.99, clip=2.0, scale=0.5)),
                ("ema", dict(gamma=0.999, clip=-1, mode="ema"))):
    w = ref_wrappers.VecNormalizeRewardWrapper(ScriptedVecEnv(), **kw)
    w.reset()
    scaled, stds, clips = [], [], []
    for t in range(T):
        _, r, _, infos = w.step(actions[t])
        scaled.append(np.asarray(r))
        stds.append(w.std)
        clips.append(int(infos[0].get("reward_clips", 0)))
    out[f"norm_{tag}_rewards"] = np.stack(scaled)
    out[f"norm_{tag}_std"] = np.asarray(stds, np.float64)
    out[f"norm_{tag}_clips"] = np.asarray(clips)
    out[f"norm_{tag}_final"] = np.asarray([w.ret_rms.mean, w.ret_rms.var, w.ret_rms.count, w.ret_var], np.float64)
    out[f"norm_{tag}_current_returns"] = np.asarray(w.current_returns)

w = ref_wrappers.VecRepeatedActionPenalty(ScriptedVecEnv(), max_repeated_actions=5, penalty=0.25)
w.reset()
pen, max_rep, flagged = [], [], []
for t in range(T):
    _, r, _, infos = w.step(actions[t])
    pen.append(np.asarray(r))
    max_rep.append(int(infos[0]["max_repeats"]))
    flagged.append([int("repeated_action" in i) for i in infos])
out["penalty_rewards"] = np.stack(pen)
out["penalty_max_repeats"] = np.asarray(max_rep)
out["penalty_flagged"] = np.asarray(flagged)

# RunningMeanStd: a stream of batches, and update_from_moments with array shapes
rms = ref_utils.RunningMeanStd(shape=(3,))
batches = [rng.standard_normal((n, 3)) * s + m for n, s, m in ((5, 1.0, 0.0), (1, 3.0, 2.0), (40, 0.1, -5.0), (7, 10.0, 1.0))]
trace = []
for b in batches:
    rms.update(b)
    trace.append(np.concatenate([rms.mean, rms.var, [rms.count]]))
out["rms_batches"] = np.concatenate(batches)
out["rms_batch_sizes"] = np.asarray([len(b) for b in batches])
out["rms_trace"] = np.stack(trace)

np.savez_compressed(os.path.join(HERE, "wrappers_golden.npz"), **out)
print("wrote wrappers_golden.npz:", {k: v.shape for k, v in out.items()})
