"""CPU: the env-family make stacks (ppo_amd/atari.py, procgen.py, mujoco.py over ppo_amd/env_wrappers.py) against the
REFERENCE's own rl.atari.make / rl.procgen.make / rl.mujoco.make run over the same scripted simulators
(tests/golden/make_env_stack_golden.py, fake_envs.py): wrapper order, every observation (sums of every step, full
frames at sampled steps and around resets), rewards, dones and the info fields the trainer reads — bit for bit,
including the np.random draws of NoopReset / FrameSkip / RandomTermination.  OpenCV is installed on neither side:
the resize inside AtariWrapper is the shared nearest-neighbour stand-in, so OpenCV's arithmetic is NOT pinned here."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
import fake_envs  # noqa: E402
from make_env_stack_golden import CONFIGS, KEYS, scripted_actions  # noqa: E402 (data + the action script only)

from ppo_amd import atari, env_wrappers, mujoco, procgen  # noqa: E402
from ppo_amd.config import args  # noqa: E402


@pytest.fixture(scope="module")
def gold():
    return (np.load(os.path.join(GOLDEN, "env_stack_golden.npz")),
            json.load(open(os.path.join(GOLDEN, "env_stack_golden.json"))))


@pytest.mark.parametrize("name", list(CONFIGS))
def test_make_stack_reproduces_the_reference_trace(name, gold, monkeypatch):
    g, meta = gold
    cfg = CONFIGS[name]
    args.setup(cfg["flags"])
    monkeypatch.setattr(env_wrappers, "resize_area", fake_envs.standin_resize)
    base = {"atari": fake_envs.FakeAtari, "procgen": fake_envs.FakeProcgen, "mujoco": fake_envs.FakeMujoco}[cfg["family"]]()
    module = {"atari": atari, "procgen": procgen, "mujoco": mujoco}[cfg["family"]]
    np.random.seed(cfg["np_seed"])
    env = module.make(cfg["env"], seed=cfg["seed"], base_env=base)
    chain, e = [], env
    while e is not None and e is not base:
        chain.append(type(e).__name__)
        e = e.__dict__.get("env")
    assert chain == meta[name]["chain"], "wrapper order differs from the reference's make()"
    assert getattr(base, "seeds", None) == meta[name]["seeds_seen"]
    np.random.seed(cfg["np_seed"] + 1)
    got = fake_envs.drive(env, scripted_actions(name, cfg), KEYS)
    want = {k[len(name) + 2:]: g[k] for k in g.files if k.startswith(name + "__")}
    assert set(got) == set(want), set(got) ^ set(want)
    for k, v in want.items():
        a = got[k]
        assert a.shape == v.shape and a.dtype == v.dtype, (k, a.shape, v.shape, a.dtype, v.dtype)
        assert np.array_equal(a, v, equal_nan=v.dtype.kind == "f"), k
    assert int(want["done"].sum()) >= 3 and np.isfinite(want["info_ep_score"]).any()


def test_numpy_image_ops_without_opencv():
    """The NumPy restatements used when cv2 is absent (parity with OpenCV unpinned: it is not installed here)."""
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (210, 160, 3), dtype=np.uint8)
    small = env_wrappers.resize_area(img, 84, 84)
    assert small.shape == (84, 84, 3) and small.dtype == np.uint8
    assert abs(float(small.mean()) - float(img.mean())) < 1.0  # area averaging preserves the mean
    assert np.array_equal(env_wrappers.resize_area(np.full((210, 160), 77, np.uint8), 84, 84), np.full((84, 84), 77, np.uint8))
    w = env_wrappers._area_weights(210, 84)
    assert np.allclose(w.sum(1), 1.0, atol=1e-6)
    gray = env_wrappers.rgb_to_gray(img)
    ref = 0.299 * img[..., 0] + 0.587 * img[..., 1] + 0.114 * img[..., 2]
    assert gray.dtype == np.uint8 and np.abs(gray.astype(np.float64) - ref).max() <= 1.0
    yuv = env_wrappers.rgb_to_yuv(img)
    assert np.array_equal(yuv[..., 0], gray) and yuv.shape == img.shape
    grey_px = np.full((2, 2, 3), 100, np.uint8)
    assert np.array_equal(env_wrappers.rgb_to_yuv(grey_px), np.stack([grey_px[..., 0], 128 + 0 * grey_px[..., 0], 128 + 0 * grey_px[..., 0]], -1))


def test_env_family_flags_follow_the_reference_defaults():
    args.setup(["--env_type=atari"])
    assert (args.env.frame_skip, args.env.frame_stack, args.env.color_mode, args.env.timeout) == (4, 4, "bw", 27000)
    args.setup(["--env_type=procgen", "--env_name=bigfish"])
    assert (args.env.frame_skip, args.env.frame_stack, args.env.color_mode, args.env.timeout) == (1, 1, "yuv", 6000)
    args.setup(["--env_type=mujoco", "--env_name=Reacher"])
    assert args.env.timeout == 51
    with pytest.raises(ValueError):
        args.setup(["--env_type=procgen", "--env_frame_stack=4"])
    args.setup([])
