"""Small built-in environments with the classic gym API (`reset() -> obs`, `step(a) -> obs, reward, done,
info`, `seed(s)`), so the process-pool vector env and the MLP policy path can be exercised end to end in
an image without gym (BASELINE.json configs[0]: "CartPole-v1, 8 envs, MLP policy").

`CartPoleEnv` integrates the cart-pole equations of Barto, Sutton & Anderson (1983) with the usual
constants of the CartPole-v1 task (force 10 N, 0.02 s Euler steps, failure at |x| > 2.4 or |theta| > 12 deg,
500-step limit, reward 1 per step).  `EpisodeInfo` adds the `time / ep_length / ep_score` info fields the
trainer logs (reference: rl/wrappers.py EpisodeScoreWrapper, TimeAwareWrapper)."""
import math

import numpy as np


class Box:
    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)


class Discrete:
    def __init__(self, n):
        self.n = int(n)


class CartPoleEnv:
    GRAVITY, M_CART, M_POLE, HALF_LEN, FORCE, TAU = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    X_LIMIT, THETA_LIMIT, MAX_STEPS = 2.4, 12 * 2 * math.pi / 360, 500

    def __init__(self, seed=None):
        self.observation_space = Box((4,), np.float32)
        self.action_space = Discrete(2)
        self._rng = np.random.default_rng(seed)
        self._s = np.zeros(4, np.float64)
        self._t = 0

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def reset(self):
        self._s = self._rng.uniform(-0.05, 0.05, size=4)
        self._t = 0
        return self._s.astype(np.float32)

    def step(self, action):
        x, x_dot, th, th_dot = self._s
        f = self.FORCE if int(action) == 1 else -self.FORCE
        m_total = self.M_CART + self.M_POLE
        pml = self.M_POLE * self.HALF_LEN
        c, s = math.cos(th), math.sin(th)
        tmp = (f + pml * th_dot * th_dot * s) / m_total
        th_acc = (self.GRAVITY * s - c * tmp) / (self.HALF_LEN * (4.0 / 3.0 - self.M_POLE * c * c / m_total))
        x_acc = tmp - pml * th_acc * c / m_total
        self._s = np.array([x + self.TAU * x_dot, x_dot + self.TAU * x_acc, th + self.TAU * th_dot,
                            th_dot + self.TAU * th_acc])
        self._t += 1
        fell = abs(self._s[0]) > self.X_LIMIT or abs(self._s[2]) > self.THETA_LIMIT
        done = bool(fell or self._t >= self.MAX_STEPS)
        return self._s.astype(np.float32), 1.0, done, {}

    def save_state(self, buffer):
        buffer["cartpole"] = (self._s.copy(), self._t, self._rng.bit_generator.state)

    def restore_state(self, buffer):
        s, self._t, rng_state = buffer["cartpole"]
        self._s = np.array(s)
        self._rng.bit_generator.state = rng_state

    def close(self):
        pass


class EpisodeInfo:
    """Adds `time` (steps into the episode), `ep_length`, `ep_score` to every info dict."""

    def __init__(self, env):
        self.env = env
        self.observation_space, self.action_space = env.observation_space, env.action_space
        self._len, self._score = 0, 0.0

    def __getattr__(self, name):
        if name == "env":
            raise AttributeError(name)
        return getattr(self.env, name)

    def reset(self):
        self._len, self._score = 0, 0.0
        return self.env.reset()

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        self._len += 1
        self._score += float(reward)
        info = dict(info, time=self._len, ep_length=self._len, ep_score=self._score)
        return obs, reward, done, info

    def save_state(self, buffer):
        buffer["episode_info"] = (self._len, self._score)
        self.env.save_state(buffer)

    def restore_state(self, buffer):
        self._len, self._score = buffer["episode_info"]
        self.env.restore_state(buffer)


def make_cartpole(seed=None):
    return EpisodeInfo(CartPoleEnv(seed))
