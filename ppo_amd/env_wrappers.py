"""Per-environment wrappers behind the env-family make functions — host mirror of the gym wrappers the reference
chains in rl/atari.py:163-230, rl/procgen.py:64-84 and rl/mujoco.py:46-67 (classes in rl/wrappers.py).  They sit
between a simulator (ALE, procgen, mujoco: not part of this build) and the process-pool vector env, on host cores,
one instance per env.

gym is not required: `Wrapper` is the small part of gym.Wrapper these classes rely on (attribute forwarding,
`unwrapped`), so a stack can be built over anything with the classic gym API — `reset() -> obs`,
`step(a) -> (obs, reward, done, info)`, `observation_space.shape/.dtype`, `action_space` — and is pinned against
the reference's own classes driven over the same scripted base env (tests/golden/make_env_stack_golden.py).
Every wrapper draws from `np.random` exactly where the reference does (NoopReset, FrameSkip, RandomTermination), so
a seeded trace is reproduced bit for bit.  Image operations the reference delegates to OpenCV (resize, colour
conversion) use cv2 when it is importable and NumPy restatements otherwise (parity of those restatements is
unpinned: cv2 is not installed in the build image).
"""
import collections
import math

import numpy as np


class Box:
    """Shape / dtype description of an observation or continuous action (what the wrappers read of gym.spaces.Box)."""

    def __init__(self, low=0, high=255, shape=(), dtype=np.uint8):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)


class Discrete:
    def __init__(self, n):
        self.n = int(n)


class Wrapper:
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):  # only reached for attributes the wrapper itself lacks
        if name.startswith("_") or name == "env":
            raise AttributeError(name)
        return getattr(self.__dict__["env"], name)

    @property
    def unwrapped(self):
        env = self.__dict__["env"]
        return env.unwrapped if hasattr(env, "unwrapped") else env

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def step(self, action):
        return self.env.step(action)

    def close(self):
        if hasattr(self.env, "close"):
            return self.env.close()


def get_wrapper(env, wrapper_type):
    """First wrapper of the given type down the `.env` chain, or None (rl/wrappers.py:1745)."""
    while env is not None:
        if isinstance(env, wrapper_type):
            return env
        env = env.__dict__.get("env") if hasattr(env, "__dict__") else None
    return None


# ---------------------------------------------------------------------------------------------- bookkeeping
class LabelEnvWrapper(Wrapper):
    """info[label_name] = label_value on every step (rl/wrappers.py:712-722)."""

    def __init__(self, env, label_name, label_value):
        super().__init__(env)
        self.label_name, self.label_value = label_name, label_value

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        info[self.label_name] = self.label_value
        return obs, reward, done, info


class MonitorWrapper(Wrapper):
    """Keeps the unmodified reward (and optionally frame) in info (rl/wrappers.py:1069-1084)."""

    def __init__(self, env, monitor_video=False):
        super().__init__(env)
        self.monitor_video = monitor_video

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        if self.monitor_video:
            info["monitor_obs"] = obs.copy()
        info["raw_reward"] = reward
        return obs, reward, done, info


class EpisodeScoreWrapper(Wrapper):
    """info["ep_score"], info["ep_length"]: running totals of the current episode (rl/wrappers.py:1421-1451)."""

    def __init__(self, env):
        super().__init__(env)
        self.ep_score, self.ep_length = 0, 0

    def reset(self, **kwargs):
        obs = self.env.reset(**kwargs)
        self.ep_score, self.ep_length = 0, 0
        return obs

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        self.ep_score += reward
        self.ep_length += 1
        info["ep_score"], info["ep_length"] = self.ep_score, self.ep_length
        return obs, reward, done, info

    def save_state(self, buffer):
        buffer["ep_score"], buffer["ep_length"] = self.ep_score, self.ep_length

    def restore_state(self, buffer):
        self.ep_score, self.ep_length = buffer["ep_score"], buffer["ep_length"]


class NullActionWrapper(Wrapper):
    """A negative action does not step the env: the last observation and info come back with reward 0, done False
    (rl/wrappers.py:1393-1418) — how a vector env skips single envs."""

    def __init__(self, env):
        super().__init__(env)
        self._prev_obs, self._prev_info = None, {}

    def reset(self, **kwargs):
        self._prev_obs = self.env.reset(**kwargs)
        return self._prev_obs

    def step(self, action):
        if action < 0:
            return self._prev_obs, 0, False, self._prev_info
        result = self.env.step(action)
        self._prev_obs, self._prev_info = result[0], result[3]
        return result


# ---------------------------------------------------------------------------------------------- episode structure
class RandomTerminationWrapper(Wrapper):
    """Ends the episode with probability p per step (rl/wrappers.py:697-710)."""

    def __init__(self, env, p: float):
        super().__init__(env)
        self.p = p

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        done = done or (np.random.rand() < self.p)  # drawn only while the env itself is not done, as the reference's `or`
        return obs, reward, done, info


class TimeLimitWrapper(Wrapper):
    """Truncates at max_episode_steps; info["time"], info["time_frac"] describe the state landed in, 0 after a done
    (rl/wrappers.py:1100-1130)."""

    def __init__(self, env, max_episode_steps=None):
        super().__init__(env)
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = 0

    def reset(self, **kwargs):
        self._elapsed_steps = 0
        return self.env.reset(**kwargs)

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            done = True
            info["TimeLimit.truncated"] = True
        info["time_frac"] = 0 if done else self._elapsed_steps / self._max_episode_steps
        info["time"] = 0 if done else self._elapsed_steps
        return obs, reward, done, info

    def save_state(self, buffer):
        buffer["_elapsed_steps"] = self._elapsed_steps

    def restore_state(self, buffer):
        self._elapsed_steps = buffer["_elapsed_steps"]


class NoopResetWrapper(Wrapper):
    """1..noop_max no-op actions (action 0) after every reset (rl/wrappers.py:1453-1501); the count is reported once
    as info["noop_start"]."""

    def __init__(self, env, noop_max=30):
        super().__init__(env)
        self.noop_max = noop_max
        self.override_num_noops = None
        self.noop_action = 0
        self.noop_given = None
        assert env.unwrapped.get_action_meanings()[0] == "NOOP"

    def reset(self, **kwargs):
        obs = self.env.reset(**kwargs)
        noops = self.override_num_noops if self.override_num_noops is not None else np.random.randint(1, self.noop_max + 1)
        assert noops >= 0
        self.noop_given = noops
        for _ in range(noops):
            obs, _, done, _ = self.env.step(self.noop_action)
            if done:
                obs = self.env.reset(**kwargs)
        return obs

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        if self.noop_given is not None:
            info["noop_start"], self.noop_given = self.noop_given, None
        return obs, reward, done, info


class FrameSkipWrapper(Wrapper):
    """Repeats the action `skip` times, sums the rewards and reduces (max) over the last two raw frames
    (rl/wrappers.py:381-455).  info["time"] counts agent interactions since reset; the simulator's own count moves to
    info["time_raw"].  A done step returns a blank frame (the vector env resets and shows the next episode's first)."""

    def __init__(self, env, min_skip=4, max_skip=None, reduce_op=np.max):
        super().__init__(env)
        max_skip = min_skip if max_skip is None else max_skip
        assert env.observation_space.dtype == "uint8"
        assert 1 <= min_skip <= max_skip
        self._obs_buffer = np.zeros((2,) + tuple(env.observation_space.shape), dtype=np.uint8)
        self._min_skip, self._max_skip, self._reduce_op = min_skip, max_skip, reduce_op
        self._t = 0

    def reset(self, **kwargs):
        self._t = 0
        return self.env.reset(**kwargs)

    def step(self, action):
        skip = np.random.randint(self._min_skip, self._max_skip + 1)  # drawn even for a fixed skip, like the reference
        total_reward, done, info = 0.0, None, {}
        for i in range(skip):
            obs, reward, done, step_info = self.env.step(action)
            if i >= skip - 2:
                self._obs_buffer[i - (skip - 2)] = obs
            if step_info is not None:
                info.update(step_info)
            total_reward += reward
            if done:
                break
        self._t += 1
        if done:
            frame = self._reduce_op(self._obs_buffer * 0, axis=0)
            self._t = 0
        else:
            frame = self._reduce_op(self._obs_buffer, axis=0)
        if "time" in info:
            info["time_raw"] = info["time"]
        info["time"] = self._t
        return frame, total_reward, done, info

    def save_state(self, buffer):
        buffer["t"] = self._t

    def restore_state(self, buffer):
        self._t = buffer["t"]


class EpisodicLifeEnv(Wrapper):
    """Loss of a life ends the episode for the learner (info["fake_done"]); the game is only reset at a real game
    over, otherwise `reset` advances one no-op step (rl/wrappers.py:344-379)."""

    def __init__(self, env):
        super().__init__(env)
        self.lives = 0
        self.was_real_done = True

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        self.was_real_done = done
        lives = self.env.unwrapped.ale.lives()
        if 0 < lives < self.lives:  # lives == 0 lingers for a few frames in some games: wait for the env's own done
            done = True
            info["fake_done"] = True
        self.lives = lives
        return obs, reward, done, info

    def reset(self, **kwargs):
        if self.was_real_done:
            obs = self.env.reset(**kwargs)
        else:
            obs, _, _, _ = self.env.step(0)
        self.lives = self.env.unwrapped.ale.lives()
        return obs


class SaveEnvStateWrapper(Wrapper):
    """Simulator state in checkpoints: ALE's clone_state / restore_state (rl/wrappers.py:516-534)."""

    def __init__(self, env, determanistic: bool = True):
        super().__init__(env)
        self.determanistic = determanistic

    def save_state(self, buffer):
        sim = self.unwrapped
        if not hasattr(sim, "clone_state"):
            raise AssertionError("Only Atari is supported for state saving/loading")
        buffer["atari"] = sim.clone_state(include_rng=self.determanistic)

    def restore_state(self, buffer):
        assert "atari" in buffer, "No state information found for Atari."
        self.unwrapped.restore_state(buffer["atari"])


class MontezumaInfoWrapper(Wrapper):
    """info["room_count"] from RAM byte 3 (rl/wrappers.py:1563-1593)."""

    def __init__(self, env, room_address=3):
        super().__init__(env)
        self.room_address = room_address
        self.visited_rooms = set()

    def get_current_room(self):
        ram = self.env.unwrapped.ale.getRAM()
        assert len(ram) == 128
        return int(ram[self.room_address])

    def reset(self, **kwargs):
        self.visited_rooms.clear()
        return self.env.reset()

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        self.visited_rooms.add(self.get_current_room())
        info["room_count"] = len(self.visited_rooms)
        if done:
            info.setdefault("episode", {}).update(visited_rooms=self.visited_rooms.copy())
        return obs, reward, done, info


# ---------------------------------------------------------------------------------------------- rewards
class ClipRewardWrapper(Wrapper):
    """Clips the reward to [-clip, clip]; the original goes to info["unclipped_reward"] (rl/wrappers.py:457-471)."""

    def __init__(self, env, clip: float):
        super().__init__(env)
        self.clip = clip

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        if reward > self.clip or reward < -self.clip:
            info["unclipped_reward"] = reward
            reward = np.clip(reward, -self.clip, +self.clip)
        return obs, reward, done, info


class SqrtRewardWrapper(Wrapper):
    """sign(r) (sqrt(|r| + 1) - 1) + epsilon r (rl/wrappers.py:536-547)."""

    def __init__(self, env, epsilon: float = 1e-3):
        super().__init__(env)
        self.epsilon = epsilon

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        sign = -1 if reward < 0 else +1
        return obs, sign * (math.sqrt(abs(reward) + 1) - 1) + self.epsilon * reward, done, info


class DeferredRewardWrapper(Wrapper):
    """Rewards are withheld and paid as one sum at step `time_limit`, or at the terminal step for -1
    (rl/wrappers.py:474-513)."""

    def __init__(self, env, time_limit=-1):
        super().__init__(env)
        self.t, self.episode_reward, self.time_limit = 0, 0, time_limit

    def reset(self):
        obs = self.env.reset()
        self.t, self.episode_reward = 0, 0
        return obs

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        self.t += 1
        pay = self.t == self.time_limit or (self.time_limit == -1 and done)
        self.episode_reward += reward
        paid = 0
        if pay:
            paid, self.episode_reward = self.episode_reward, 0
        return obs, paid, done, info

    def save_state(self, buffer):
        buffer["t"], buffer["episode_reward"] = self.t, self.episode_reward

    def restore_state(self, buffer):
        self.t, self.episode_reward = buffer["t"], buffer["episode_reward"]


# ---------------------------------------------------------------------------------------------- observations
def _area_weights(n_in: int, n_out: int) -> np.ndarray:
    """[n_out, n_in] coverage weights of area interpolation: output cell o averages input cells over
    [o * s, (o + 1) * s), s = n_in / n_out, partially covered cells in proportion."""
    s = n_in / n_out
    w = np.zeros((n_out, n_in), np.float32)
    for o in range(n_out):
        lo, hi = o * s, (o + 1) * s
        for i in range(int(math.floor(lo)), min(int(math.ceil(hi)), n_in)):
            w[o, i] = max(0.0, min(hi, i + 1) - max(lo, i)) / s
    return w


def resize_area(img: np.ndarray, height: int, width: int) -> np.ndarray:
    """uint8 [H, W] or [H, W, C] -> [height, width(, C)] by area averaging (what the reference asks of
    cv2.resize(..., interpolation=cv2.INTER_AREA)); cv2 itself when it is importable."""
    try:
        import cv2
        return cv2.resize(img, (width, height), interpolation=cv2.INTER_AREA)
    except ImportError:
        pass
    wy, wx = _area_weights(img.shape[0], height), _area_weights(img.shape[1], width)
    x = img.astype(np.float32)
    out = np.einsum("oh,hw...->ow...", wy, x)
    out = np.einsum("pw,ow...->op...", wx, out)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def rgb_to_gray(img: np.ndarray) -> np.ndarray:
    """[H, W, 3] uint8 -> [H, W] uint8, OpenCV's fixed-point luma (0.299, 0.587, 0.114 at 14 bits, rounded)."""
    try:
        import cv2
        return cv2.cvtColor(img, cv2.COLOR_RGB2GRAY)
    except ImportError:
        pass
    x = img.astype(np.int32)
    return ((x[..., 0] * 4899 + x[..., 1] * 9617 + x[..., 2] * 1868 + (1 << 13)) >> 14).astype(np.uint8)


def rgb_to_yuv(img: np.ndarray) -> np.ndarray:
    """[H, W, 3] uint8 RGB -> YUV, OpenCV's 8-bit fixed-point form (U = 0.492 (B - Y) + 128, V = 0.877 (R - Y) + 128)."""
    try:
        import cv2
        return cv2.cvtColor(img, cv2.COLOR_RGB2YUV)
    except ImportError:
        pass
    x = img.astype(np.int32)
    r, g, b = x[..., 0], x[..., 1], x[..., 2]
    y = (r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14
    u = ((b - y) * 8061 + (128 << 14) + (1 << 13)) >> 14
    v = ((r - y) * 14369 + (128 << 14) + (1 << 13)) >> 14
    return np.clip(np.stack([y, u, v], axis=-1), 0, 255).astype(np.uint8)


class AtariWrapper(Wrapper):
    """210x160(x3) uint8 frames -> width x height x C frames, HWC (rl/wrappers.py:1133-1200; the reference passes
    (width, height) = (res_x, res_y) and produces arrays of shape (width, height, C))."""

    def __init__(self, env, width=84, height=84):
        super().__init__(env)
        self._width, self._height = width, height
        assert env.observation_space.dtype == np.uint8, "Invalid dtype {}".format(env.observation_space.dtype)
        assert tuple(env.observation_space.shape) in [(210, 160), (210, 160, 3)], "Invalid shape {}".format(env.observation_space.shape)
        self.n_channels = 3
        self.observation_space = Box(0, 255, (self._width, self._height, self.n_channels), np.uint8)

    def _process_frame(self, obs):
        assert len(obs.shape) in [2, 3]
        if obs.ndim == 2:
            obs = obs[:, :, None]
        if obs.shape[:2] != (self._width, self._height):
            obs = resize_area(obs, self._width, self._height)
        return obs[:, :, None] if obs.ndim == 2 else obs

    def reset(self):
        return self._process_frame(self.env.reset())

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        info["channels"] = ["ColorR", "ColorG", "ColorB"]
        return self._process_frame(obs), reward, done, info


class ZeroObsWrapper(Wrapper):
    """Blank observations, a debugging aid (rl/wrappers.py:724-734)."""

    def reset(self, **kwargs):
        return self.env.reset(**kwargs) * 0

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        return obs * 0, reward, done, info


class ColorTransformWrapper(Wrapper):
    """HWC RGB -> bw (1 channel) / rgb / yuv / hsv; single-channel input passes through (rl/wrappers.py:1291-1354)."""
    NAMES = {"bw": ["Gray"], "rgb": ["ColorR", "ColorG", "ColorB"], "yuv": ["ColorY", "ColorU", "ColorV"],
             "hsv": ["ColorH", "ColorS", "ColorV"]}

    def __init__(self, env, color_mode: str):
        super().__init__(env)
        H, W, C = env.observation_space.shape
        assert C < H, f"Input should be in HWC format, not CHW, shape was {env.observation_space.shape}"
        assert color_mode in self.NAMES, f'Color mode should be one of ["bw", "rgb", "yuv", "hsv"] but was {color_mode}'
        self.expected_input_shape = (H, W, C)
        if color_mode == "bw":
            assert C in [1, 3]
        else:
            assert C == 3, f"Expecting 3 channels, found {C}"
        self.color_mode = color_mode
        self.observation_space = Box(0, 255, (H, W, 1 if color_mode == "bw" else 3), np.uint8)

    def _process_frame(self, obs):
        assert obs.shape == self.expected_input_shape, f"Shape missmatch, expecting {self.expected_input_shape} found {obs.shape}"
        if obs.shape[2] == 1 or self.color_mode == "rgb":
            return obs
        if self.color_mode == "bw":
            return rgb_to_gray(obs)[:, :, None]
        if self.color_mode == "yuv":
            return rgb_to_yuv(obs)
        import cv2  # hsv: OpenCV only
        return cv2.cvtColor(obs, cv2.COLOR_RGB2HSV)

    def reset(self):
        return self._process_frame(self.env.reset())

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        info["channels"] = list(self.NAMES[self.color_mode])
        return self._process_frame(obs), reward, done, info


class ActionAwareWrapper(Wrapper):
    """Marks the action that led to a frame as a white 4x4 block on it (rl/wrappers.py:109-150): rows
    [4a, 4a + 4) of the first four columns for HW / HWC frames, of every channel for CHW frames."""
    BLOCK = 4

    def _mark(self, obs, action):
        assert obs.dtype == np.uint8
        if action >= 0:
            lo, hi = action * self.BLOCK, action * self.BLOCK + self.BLOCK
            if obs.ndim == 2:
                obs[lo:hi, 0:self.BLOCK] = 255
            elif obs.shape[0] < obs.shape[1]:  # C, H, W
                obs[:, lo:hi, 0:self.BLOCK] = 255
            else:                              # H, W, C
                obs[lo:hi, 0:self.BLOCK, :] = 255
        return obs

    def reset(self, **kwargs):
        return self._mark(self.env.reset(**kwargs), -1)

    def step(self, action):
        assert isinstance(action, (int, np.integer)), f"Action aware requires discrete actions, but found action of type {type(action)}"
        obs, reward, done, info = self.env.step(action)
        return self._mark(obs, action), reward, done, info


class FrameStack(Wrapper):
    """The last n frames concatenated on the channel axis, newest first; a reset fills the stack with the first frame
    (rl/wrappers.py:1503-1561).  HWC uint8 in and out."""

    def __init__(self, env, n_stacks=4):
        super().__init__(env)
        assert len(env.observation_space.shape) == 3, "Invalid shape {}".format(env.observation_space.shape)
        assert env.observation_space.dtype == np.uint8, "Invalid dtype {}".format(env.observation_space.dtype)
        h, w, c = env.observation_space.shape
        assert c < h, "Must have channels first."
        self.n_stacks, self.original_channels, self.n_channels = n_stacks, c, n_stacks * c
        self.stack = collections.deque([np.zeros((h, w, c), np.uint8) for _ in range(n_stacks)], maxlen=n_stacks)
        self.observation_space = Box(0, 255, (h, w, self.n_channels), np.uint8)

    def get_obs(self):
        return np.concatenate(self.stack, axis=-1)

    def reset(self):
        obs = self.env.reset()
        for _ in range(self.n_stacks):
            self.stack.appendleft(obs)
        return self.get_obs()

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        self.stack.appendleft(obs)
        if "channels" in info:
            info["channels"] = info["channels"] * self.n_stacks
        return self.get_obs(), reward, done, info

    def save_state(self, buffer):
        buffer["stack"] = self.stack

    def restore_state(self, buffer):
        self.stack = buffer["stack"]


class TimeChannelWrapper(Wrapper):
    """Appends a channel holding uint8(255 * time_frac) (rl/wrappers.py:1235-1267); needs TimeLimitWrapper below."""

    def __init__(self, env):
        super().__init__(env)
        H, W, C = env.observation_space.shape
        assert C < H, f"Input should be in HWC format, not CHW, shape was {env.observation_space.shape}"
        self.observation_space = Box(0, 255, (H, W, C + 1), np.uint8)

    @staticmethod
    def _with_time(obs, time):
        assert obs.dtype == np.uint8
        H, W, C = obs.shape
        assert C < H, "Must be channels first."
        out = np.zeros((H, W, C + 1), np.uint8)
        out[:, :, :-1] = obs
        out[:, :, -1] = time * 255
        return out

    def reset(self):
        return self._with_time(self.env.reset(), 0)

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        assert "time_frac" in info, "must include timelimit wrapper before TimeChannelWrapper"
        if "channels" in info:
            info["channels"] += ["Gray"]
        return self._with_time(obs, info["time_frac"]), reward, done, info


class TimeFeatureWrapper(Wrapper):
    """Appends time_frac as one more feature of a flat observation (rl/wrappers.py:1203-1232)."""

    def __init__(self, env):
        super().__init__(env)
        shape = env.observation_space.shape
        assert len(shape) == 1, f"Input should in R^D, shape was {shape}"
        self.observation_space = Box(0, 255, (shape[0] + 1,), env.observation_space.dtype)

    @staticmethod
    def _with_time(obs, time):
        out = np.zeros((obs.shape[0] + 1,), dtype=obs.dtype)
        out[:-1] = obs
        out[-1] = time
        return out

    def reset(self):
        return self._with_time(self.env.reset(), 0)

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        assert "time_frac" in info, "must include timelimit wrapper before TimeChannelWrapper"
        return self._with_time(obs, info["time_frac"]), reward, done, info


class StateHistoryWrapper(Wrapper):
    """Draws a 7x7-compressed history of earlier frames onto channel 0 (rl/wrappers.py:241-291), CHW uint8."""

    def __init__(self, env):
        super().__init__(env)
        self.state_history = collections.deque(maxlen=100)

    def compressed_state(self, x):
        small = resize_area(x[-1], 7, 7)
        assert small.dtype == np.uint8
        return small.ravel()

    def _draw(self, obs):
        assert obs.dtype == np.uint8
        W = obs.shape[-1]
        n_actions = self.action_space.n
        obs[0, n_actions:n_actions + 49, :] = 0
        for x, state in enumerate(list(self.state_history)[:W]):
            obs[0, n_actions:n_actions + 49, x] = state
        return obs

    def reset(self, **kwargs):
        obs = self.env.reset(**kwargs)
        self.state_history.clear()  # (the reference clears a non-existent `action_history` here and would raise)
        return self._draw(obs)

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        self.state_history.appendleft(self.compressed_state(obs))
        return self._draw(obs), reward, done, info

    def save_state(self, buffer):
        buffer["state_history"] = self.state_history

    def restore_state(self, buffer):
        self.state_history = buffer["state_history"]


class ChannelsFirstWrapper(Wrapper):
    """HWC -> CHW (rl/wrappers.py:1269-1289)."""

    def __init__(self, env):
        super().__init__(env)
        H, W, C = env.observation_space.shape
        assert C < H, f"Input should be in HWC format, not CHW, shape was {env.observation_space.shape}"
        self.observation_space = Box(0, 255, (C, H, W), np.uint8)

    def reset(self):
        return self.env.reset().transpose(2, 0, 1)

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        return obs.transpose(2, 0, 1), reward, done, info


class F32Wrapper(Wrapper):
    """float64 simulator states -> float32 (rl/mujoco.py:11-26)."""

    def __init__(self, env):
        super().__init__(env)
        env.observation_space.dtype = np.dtype(np.float32)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs).astype(np.float32)

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        return obs.astype(np.float32), reward, done, info


class ProcGenWrapper(Wrapper):
    """Procgen frames are RGB already; names the channels (rl/procgen.py:17-30)."""

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        info["channels"] = ["Gray", "Gray", "Gray"]
        return obs, reward, done, info
