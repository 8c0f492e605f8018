"""GPU: the whole IMPALA path has to LEARN from pixels, end to end through the Runner — uint8 observations,
fused conv+pool, residual blocks, dense, heads, action sampling, GAE, the PPO loss, backward-data / weight
gradients, Adam.  Parity tests pin each of those to the reference; this one checks that together they do what
PPO is for.

The task is a contextual bandit a CNN can solve and a constant policy cannot: each observation shows a bright
patch at one of six places (plus noise) and the reward is 1 for the action with that index, 0 otherwise; every
step ends the episode.  A uniform policy earns 1/6."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


class PatchBanditVecEnv:
    """gym-API vector env (`reset() -> obs`, `step(actions) -> obs, rew, done, infos`)."""
    CENTRES = [(14, 14), (14, 42), (14, 70), (70, 14), (70, 42), (70, 70)]

    def __init__(self, num_envs, seed):
        self.num_envs = num_envs
        self.rng = np.random.default_rng(seed)
        self.target = np.zeros(num_envs, np.int64)

    def _draw(self):
        A = self.num_envs
        self.target = self.rng.integers(0, 6, A)
        obs = self.rng.integers(0, 40, size=(A, 4, 84, 84), dtype=np.uint8)
        for a, t in enumerate(self.target):
            cy, cx = self.CENTRES[t]
            obs[a, :, cy - 8:cy + 8, cx - 8:cx + 8] = 255
        return obs

    def reset(self):
        return self._draw()

    def step(self, actions):
        rew = (np.asarray(actions) == self.target).astype(np.float32)
        done = np.ones(self.num_envs, bool)
        infos = [{"ep_length": 1, "ep_score": float(r), "time": 0} for r in rew]
        return self._draw(), rew, done, infos

    def close(self):
        pass


def test_impala_cnn_learns_a_pixel_bandit():
    from ppo_amd import logger, models, rollout
    from ppo_amd.config import args
    args.setup(["--agents=64", "--n_steps=16", "--model_architecture=single", "--model_encoder=impala",
                "--env_type=synthetic", "--env_embed_time=False", "--seed=1", "--device=cuda",
                "--policy_opt_mini_batch_size=256", "--policy_opt_epochs=2", "--policy_opt_lr=0.0005",
                "--disable_logging=True", "--env_reward_normalization=off"])
    torch.manual_seed(1)
    np.random.seed(1)
    model = models.TVFModel("impala", input_dims=(4, 84, 84), actions=6, device="cuda", architecture="single",
                            hidden_units=256, head_scale=0.1, head_bias=True)
    r = rollout.Runner(model, logger.Logger(quiet=True))
    r.vec_env = PatchBanditVecEnv(64, seed=3)
    r.reset()
    scores = []
    for _ in range(12):
        r.generate_rollout()
        scores.append(float(r.ext_rewards.mean()))
        r.calculate_returns()
        r.train()
    early, late = scores[0], np.mean(scores[-3:])
    assert abs(early - 1 / 6) < 0.08, scores           # the first rollout (untrained policy) is at chance
    assert late > 0.8, scores                           # reads the patch position off the pixels
    assert torch.isfinite(r.net.flat).all()
    # greedy actions on fresh observations are the targets
    obs = r.vec_env.reset()
    out = model.forward(obs, output="policy", policy_temperature=1.0)
    greedy = out["log_policy"].argmax(dim=1).cpu().numpy()
    assert (greedy == r.vec_env.target).mean() > 0.9
