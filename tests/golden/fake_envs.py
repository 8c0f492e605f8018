"""Scripted stand-ins for the simulators (ALE / procgen / mujoco) with the classic gym API, shared by the fixture
generator (which wraps them in the REFERENCE's make functions) and the tests (which wrap them in ppo_amd's).
Everything is a pure function of (episode number, step in episode, action), so both sides see the same base env."""
import numpy as np


class _Space:
    def __init__(self, shape=None, dtype=None, n=None):
        self.shape, self.dtype, self.n = (tuple(shape) if shape is not None else None), (np.dtype(dtype) if dtype else None), n


def _frame(h, w, episode, t):
    """Low-entropy uint8 RGB frame: a moving block, a stripe whose level follows t, a corner tag of the episode."""
    f = np.zeros((h, w, 3), np.uint8)
    y, x = (7 * t + 3 * episode) % (h - 12), (11 * t + 5 * episode) % (w - 12)
    f[y:y + 12, x:x + 12, 0] = (40 + 9 * t) % 256
    f[y:y + 12, x:x + 12, 1] = (200 - 5 * t) % 256
    f[(3 * t) % h, :, 2] = (t * 13 + 17 * episode) % 256
    f[-6:, -6:, :] = (31 * episode + 1) % 256
    return f


class _Ale:
    def __init__(self, env):
        self._env = env

    def lives(self):
        return self._env._lives

    def getRAM(self):
        ram = np.zeros(128, np.uint8)
        ram[3] = (self._env._t // 9) % 5
        return ram


class FakeAtari:
    """210 x 160 x 3 frames; episode e lasts LENGTHS[e % 4] raw frames; a life is lost every 37 raw frames."""
    LENGTHS = (150, 61, 233, 97)
    REWARDS = (0.0, 1.0, -3.0, 0.0, 7.5, 0.0, 0.25, -0.5)

    def __init__(self, h=210, w=160):
        self.observation_space = _Space((h, w, 3), np.uint8)
        self.action_space = _Space(n=6)
        self.ale = _Ale(self)
        self._h, self._w = h, w
        self._episode, self._t, self._lives, self.seeds = -1, 0, 3, []

    @property
    def unwrapped(self):
        return self

    def get_action_meanings(self):
        return ["NOOP", "FIRE", "UP", "RIGHT", "LEFT", "DOWN"]

    def seed(self, s=None):
        self.seeds.append(s)

    def reset(self, **kwargs):
        self._episode += 1
        self._t, self._lives = 0, 3
        return _frame(self._h, self._w, self._episode, 0)

    def step(self, action):
        assert 0 <= int(action) < 6, action
        self._t += 1
        if self._t % 37 == 0 and self._lives > 0:
            self._lives -= 1
        reward = self.REWARDS[(self._t + 2 * int(action) + self._episode) % len(self.REWARDS)]
        done = self._t >= self.LENGTHS[self._episode % len(self.LENGTHS)]
        return _frame(self._h, self._w, self._episode, self._t), reward, done, {"lives": self._lives}


class FakeProcgen:
    """64 x 64 x 3 frames, 15 actions, episodes of 45 / 120 / 18 steps."""
    LENGTHS = (45, 120, 18)

    def __init__(self):
        self.observation_space = _Space((64, 64, 3), np.uint8)
        self.action_space = _Space(n=15)
        self._episode, self._t = -1, 0

    @property
    def unwrapped(self):
        return self

    def reset(self, **kwargs):
        self._episode += 1
        self._t = 0
        return _frame(64, 64, self._episode, 0)

    def step(self, action):
        assert 0 <= int(action) < 15, action
        self._t += 1
        done = self._t >= self.LENGTHS[self._episode % len(self.LENGTHS)]
        return _frame(64, 64, self._episode, self._t), (10.0 if done else 0.0), done, {"level_seed": self._episode}


class FakeMujoco:
    """11 float64 features, 3 continuous actions, episodes of 70 steps."""

    def __init__(self):
        self.observation_space = _Space((11,), np.float64)
        self.action_space = _Space((3,), np.float32)
        self._episode, self._t, self.seeds = -1, 0, []

    @property
    def unwrapped(self):
        return self

    def seed(self, s=None):
        self.seeds.append(s)

    def _obs(self):
        return np.sin(np.arange(11) * 0.37 + self._t * 0.11 + self._episode)

    def reset(self, **kwargs):
        self._episode += 1
        self._t = 0
        return self._obs()

    def step(self, action):
        self._t += 1
        reward = float(1.0 - 0.1 * np.square(np.asarray(action, np.float64)).sum() + 0.01 * self._t)
        return self._obs(), reward, self._t >= 70, {}


def standin_resize(img, rows, cols):
    """Nearest-neighbour resize used on BOTH sides of the env-stack fixture in place of OpenCV (not installed in the
    build image): the fixture pins the wrapper stack around the resize, not OpenCV's interpolation arithmetic."""
    ys = (np.arange(rows) * img.shape[0]) // rows
    xs = (np.arange(cols) * img.shape[1]) // cols
    return img[ys][:, xs]


def drive(env, actions, keys):
    """The vector-env worker's loop (reset on done) over scripted actions; returns the trace as arrays: every
    observation's plain and position-weighted sums, full observations at a few steps, rewards, dones and the named
    info entries (NaN where absent; `channels` as a joined string)."""
    trace = {"reward": [], "done": [], "obs_sum": [], "obs_wsum": [], "channels": []}
    info_rows = {k: [] for k in keys}
    full = {}
    obs = env.reset()
    full["obs_reset"] = np.asarray(obs).copy()
    for t, a in enumerate(actions):
        obs, reward, done, info = env.step(a)
        o = np.asarray(obs)
        flat = o.astype(np.float64).ravel()
        trace["obs_sum"].append(flat.sum())
        trace["obs_wsum"].append(float((flat * ((np.arange(flat.size) % 251) + 1)).sum()))
        trace["reward"].append(float(reward))
        trace["done"].append(bool(done))
        trace["channels"].append(",".join(info.get("channels", [])))
        for k in keys:
            v = info.get(k, np.nan)
            info_rows[k].append(float(v) if not isinstance(v, bool) else float(v))
        if t % 16 == 5 or done:
            full[f"obs_{t}"] = o.copy()
        if done:
            obs = env.reset()
            full[f"obs_after_reset_{t}"] = np.asarray(obs).copy()
    out = {k: np.asarray(v) for k, v in trace.items()}
    out.update({"info_" + k.replace(".", "_"): np.asarray(v, np.float64) for k, v in info_rows.items()})
    out.update(full)
    return out
