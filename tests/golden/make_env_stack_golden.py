#!/usr/bin/env python3
"""Generate tests/golden/env_stack_golden.npz by running the REFERENCE's own env construction — rl.atari.make
(rl/atari.py:119-230), rl.procgen.make (rl/procgen.py:33-84), rl.mujoco.make (rl/mujoco.py:29-67), i.e. its gym
wrappers of rl/wrappers.py chained in its order — over scripted stand-ins for the simulators (fake_envs.py; the
reference's `gym.make` call is pointed at them).  Build container only (needs /root/reference; see ref_shim.py):

    python tests/golden/make_env_stack_golden.py

OpenCV is not installed here: colour mode "rgb" avoids the conversion call, and `cv2.resize` (which AtariWrapper always
reaches: the reference's res_x and res_y both read the first element of the resolution, rl/config.py:543-548) is
replaced by fake_envs.standin_resize on both sides, so the fixture pins everything around the resize, not OpenCV.  Each configuration runs in its own interpreter (the reference's
flags are process-global class attributes)."""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

KEYS = ["time", "time_frac", "time_raw", "ep_score", "ep_length", "raw_reward", "noop_start", "fake_done",
        "unclipped_reward", "TimeLimit.truncated", "lives", "seed"]
CONFIGS = {
    "atari_a": dict(family="atari", env="FakeGame", seed=11, np_seed=5, steps=150, flags=[
        "--env_type=atari", "--env_resolution=nature", "--env_color_mode=rgb", "--env_atari_rom_check=False",
        "--env_timeout=40", "--env_noop_duration=5", "--env_reward_clipping=1"]),
    "atari_b": dict(family="atari", env="FakeGame", seed=None, np_seed=6, steps=150, flags=[
        "--env_type=atari", "--env_resolution=nature", "--env_color_mode=rgb", "--env_atari_rom_check=False",
        "--env_timeout=0", "--env_noop_duration=0", "--env_reward_clipping=sqrt",
        "--env_per_step_termination_probability=0.05", "--env_atari_terminal_on_loss_of_life=True",
        "--env_deferred_rewards=-1", "--env_embed_action=False", "--env_frame_stack=2", "--env_embed_time=False"]),
    "procgen": dict(family="procgen", env="fakerun", seed=3, np_seed=7, steps=220, flags=[
        "--env_type=procgen", "--env_color_mode=rgb", "--env_timeout=100"]),
    "mujoco": dict(family="mujoco", env="FakeWalker", seed=4, np_seed=8, steps=160, flags=[
        "--env_type=mujoco", "--env_timeout=50"]),
}


def scripted_actions(name, cfg):
    rng = np.random.default_rng(cfg["np_seed"] + 100)
    if cfg["family"] == "mujoco":
        return [rng.uniform(-1, 1, 3).astype(np.float32) for _ in range(cfg["steps"])]
    n = 15 if cfg["family"] == "procgen" else 6
    a = rng.integers(0, n, cfg["steps"])
    a[rng.random(cfg["steps"]) < 0.08] = -1  # NullActionWrapper: the vector env skipping this env
    return [int(x) for x in a]


def run_one(name):
    from ref_shim import load_reference
    import fake_envs
    cfg = CONFIGS[name]
    rl = load_reference(cfg["flags"] + ["--output_folder=/tmp/ref_golden_out"])
    import rl.atari, rl.mujoco, rl.procgen
    base = {"atari": fake_envs.FakeAtari, "procgen": fake_envs.FakeProcgen, "mujoco": fake_envs.FakeMujoco}[cfg["family"]]()
    module = {"atari": rl.atari, "procgen": rl.procgen, "mujoco": rl.mujoco}[cfg["family"]]
    made = {}

    def fake_make(env_name, **kwargs):
        made["name"], made["kwargs"] = env_name, {k: (v if isinstance(v, (int, float, str, bool)) else str(v)) for k, v in kwargs.items()}
        return base
    module.gym.make = fake_make
    sys.modules["cv2"].resize = lambda img, dsize, interpolation=None: fake_envs.standin_resize(img, dsize[1], dsize[0])
    np.random.seed(cfg["np_seed"])
    env = module.make(cfg["env"], seed=cfg["seed"]) if cfg["family"] != "procgen" else module.make(cfg["env"], seed=cfg["seed"])
    np.random.seed(cfg["np_seed"] + 1)  # make() seeds np.random with `seed`; the trace starts from a known state
    out = fake_envs.drive(env, scripted_actions(name, cfg), KEYS)
    chain, e = [], env
    while e is not None and not isinstance(e, type(base)):
        chain.append(type(e).__name__)
        e = e.__dict__.get("env")
    np.savez_compressed(f"/tmp/env_stack_{name}.npz", **out)
    json.dump({"gym_make": made, "chain": chain, "seeds_seen": getattr(base, "seeds", None)}, open(f"/tmp/env_stack_{name}.json", "w"))


def main():
    if len(sys.argv) > 1:
        return run_one(sys.argv[1])
    allout, meta = {}, {}
    for name, cfg in CONFIGS.items():
        subprocess.check_call([sys.executable, os.path.abspath(__file__), name], stderr=subprocess.DEVNULL)
        z = np.load(f"/tmp/env_stack_{name}.npz")
        for k in z.files:
            allout[f"{name}__{k}"] = z[k]
        meta[name] = dict(cfg, **json.load(open(f"/tmp/env_stack_{name}.json")))
        print(name, meta[name]["chain"], "dones", int(z["done"].sum()), "reward sum", float(z["reward"].sum()))
    np.savez_compressed(os.path.join(HERE, "env_stack_golden.npz"), **allout)
    json.dump(meta, open(os.path.join(HERE, "env_stack_golden.json"), "w"), indent=1)
    print("wrote", len(allout), "arrays;", os.path.getsize(os.path.join(HERE, "env_stack_golden.npz")) / 1e6, "MB")


if __name__ == "__main__":
    main()
