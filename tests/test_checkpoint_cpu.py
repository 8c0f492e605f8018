"""CPU: the checkpoint container (rl/rollout.py:394-517) — plain-data conversion that loads under
torch.load(weights_only=True), the gzip container and file naming, and env / wrapper state gathered and
restored through the wrapper chain and the process pool."""
import functools
import os

import numpy as np
import pytest
import torch

from ppo_amd import checkpoint, classic_envs, wrappers
from ppo_amd.hybrid_vec_env import HybridAsyncVectorEnv
from ppo_amd.ppo import get_checkpoints


def test_plain_round_trip_and_weights_only_load(tmp_path):
    rng = np.random.default_rng(3)
    data = {"step": 123456, "f": 0.25, "name": "x", "none": None, "flag": True,
            "tensor": torch.arange(5, dtype=torch.float32),
            "nd": {"f64": rng.standard_normal((3, 2)), "u64": np.array([2**63 + 5, 7], np.uint64), "b": np.array([True, False]),
                   "empty": np.zeros((0, 4), np.float32)},
            "scalar": np.float32(1.5), "tuple": (np.zeros(2), 3, ("a", 2.0)),
            "rng": rng.bit_generator.state, "list": [1, [2, 3]]}
    for compress in (True, False):
        path = checkpoint.save(data, str(tmp_path / "checkpoint-001M-params.pt"), compress)
        assert path.endswith(".pt.gz") == compress and os.path.exists(path)
        back = checkpoint.load(str(tmp_path / "checkpoint-001M-params.pt") if compress else path)
        assert back["step"] == 123456 and back["f"] == 0.25 and back["name"] == "x" and back["none"] is None
        assert torch.equal(back["tensor"], data["tensor"])
        for k, v in data["nd"].items():
            assert back["nd"][k].dtype == v.dtype and back["nd"][k].shape == v.shape and np.array_equal(back["nd"][k], v)
        assert isinstance(back["scalar"], np.float32) and back["scalar"] == np.float32(1.5)
        assert isinstance(back["tuple"], tuple) and back["tuple"][2] == ("a", 2.0) and np.array_equal(back["tuple"][0], np.zeros(2))
        assert back["rng"] == data["rng"] and back["list"] == [1, [2, 3]]
        os.remove(path)
    with pytest.raises(TypeError):
        checkpoint.to_plain({"x": object()})
    # naming as the reference's get_checkpoints expects
    for name in ("checkpoint-003M-params.pt.gz", "checkpoint-012M-params.pt", "other.pt"):
        open(tmp_path / name, "wb").close()
    assert get_checkpoints(str(tmp_path)) == [(12, "checkpoint-012M-params.pt"), (3, "checkpoint-003M-params.pt.gz")]


def test_env_state_through_wrappers_and_pool(tmp_path):
    fns = [functools.partial(classic_envs.make_cartpole, 50 + i) for i in range(4)]
    env = wrappers.VecRepeatedActionPenalty(
        wrappers.VecNormalizeRewardWrapper(HybridAsyncVectorEnv(fns, max_cpus=2), gamma=0.99), 3, 0.5)
    try:
        env.reset()
        a = np.array([0, 1, 1, 0])
        for _ in range(7):
            env.step(a)
        state = checkpoint.save_env_state(env)
        assert set(state) == {"VecNormalizeRewardWrapper", "HybridAsyncVectorEnv"}
        assert sorted(state["HybridAsyncVectorEnv"]) == [f"vec_{i:03d}" for i in range(4)]
        path = checkpoint.save({"env_state": state}, str(tmp_path / "cp.pt"), True)
        after = [env.step(a) for _ in range(5)]
        loaded = checkpoint.load(path)["env_state"]
        checkpoint.restore_env_state(env, loaded)
        again = [env.step(a) for _ in range(5)]
        for (o1, r1, d1, _), (o2, r2, d2, _) in zip(after, again):
            assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(d1, d2)
    finally:
        env.close()
