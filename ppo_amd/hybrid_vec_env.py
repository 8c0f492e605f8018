"""Process-pool vector environment for gym-API envs — host mirror of the reference's
rl/hybridVecEnv.py `HybridAsyncVectorEnv` (:49-203): `max_cpus` worker processes, each stepping
`len(env_fns) / max_cpus` envs, auto-reset on done, action -1 = leave that env untouched
(rl/wrappers.py:1393-1418), `seed`, `save_state` / `restore_state` through the workers.

MI355X-first differences: observations, rewards and dones of all envs land in ONE shared-memory block
that the workers write in place and the trainer's process registers with the HIP runtime as pinned host
memory, so the per-step upload is a single async H2D copy straight out of the block (the reference
copies worker -> shared memory -> numpy -> `obs.copy()` -> pageable H2D: rl/rollout.py:816, 2349-2372).
Only the per-env info dicts travel through the pipes.  This module imports numpy only (workers start
in ~0.1 s); torch is touched lazily, in the parent, to pin the block.
"""
import multiprocessing as mp
import os
import sys
import traceback
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def _as_array(raw, dtype, shape):
    return np.frombuffer(raw, dtype=dtype).reshape(shape)


def _worker(index, env_fns, pipe, raws, obs_shape, obs_dtype, first, total, threaded):
    try:
        os.nice(1)  # the trainer's thread keeps the GPU fed; workers yield to it (rl/hybridVecEnv.py:153)
    except OSError:
        pass
    obs = _as_array(raws[0], obs_dtype, (total, *obs_shape))
    rew = _as_array(raws[1], np.float32, (total,))
    done = _as_array(raws[2], np.uint8, (total,))
    envs, pool = [], None
    try:
        envs = [fn() for fn in env_fns]
        n = len(envs)
        last_info = [{} for _ in range(n)]
        pool = ThreadPoolExecutor(max_workers=2) if threaded and n > 1 else None

        def step_one(j, action):
            if np.ndim(action) == 0 and action < 0:  # null action: env frozen, reward 0, not done
                rew[first + j], done[first + j] = 0.0, 0
                return last_info[j]
            o, r, d, info = envs[j].step(action)
            if d:
                o = envs[j].reset()
            obs[first + j], rew[first + j], done[first + j] = o, r, d
            last_info[j] = info
            return info

        while True:
            command, data = pipe.recv()
            if command == "step":
                if pool is not None:
                    infos = list(pool.map(step_one, range(n), data))
                else:
                    infos = [step_one(j, data[j]) for j in range(n)]
                pipe.send((infos, True))
            elif command == "reset":
                for j, env in enumerate(envs):
                    obs[first + j] = env.reset()
                    last_info[j] = {}
                pipe.send((None, True))
            elif command == "seed":
                for env, s in zip(envs, data):
                    env.seed(s)
                pipe.send((None, True))
            elif command == "save":
                out = {}
                for j, env in enumerate(envs):
                    buffer = {}
                    env.save_state(buffer)
                    out[f"vec_{j:03d}"] = buffer
                pipe.send((out, True))
            elif command == "load":
                for j, env in enumerate(envs):
                    key = f"vec_{j:03d}"
                    if key in data:
                        env.restore_state(data[key])
                pipe.send((None, True))
            elif command == "close":
                pipe.send((None, True))
                break
            else:
                raise RuntimeError(f"Received unknown command `{command}`.")
    except (KeyboardInterrupt, Exception):  # report, never hang the parent
        try:
            pipe.send(("".join(traceback.format_exception(*sys.exc_info())), False))
        except (BrokenPipeError, OSError):
            pass
    finally:
        if pool is not None:
            pool.shutdown(wait=False)
        for env in envs:
            try:
                env.close()
            except Exception:
                pass


class PoolGroup:
    """A contiguous range of a HybridAsyncVectorEnv's envs (whole workers) as one array-stepping vector env."""

    def __init__(self, pool, index):
        self.pool, self.index = pool, index
        first_w, n_w = pool._group_workers(index)
        self.lo, self.hi = first_w * pool.n_sequential, (first_w + n_w) * pool.n_sequential
        self.num_envs = self.hi - self.lo
        self.obs = pool.obs[self.lo:self.hi]
        self._time = np.zeros(self.num_envs, np.int32)
        self._len = np.zeros(self.num_envs, np.int32)
        self._score = np.zeros(self.num_envs, np.float32)
        self.infos = []

    @property
    def obs_t(self):
        import torch
        return torch.from_numpy(self.obs)

    def step_arrays(self, actions, rew_out=None, done_out=None):
        """Step the group's envs; rewards / dones are copied into the given rows.  Returns (obs, rew, done) views."""
        self.pool.step_async(self.index, actions)
        self.infos = self.pool.step_wait(self.index)
        rew, done = self.pool._rew[self.lo:self.hi], self.pool._done[self.lo:self.hi]
        if rew_out is not None:
            rew_out[:] = rew
        if done_out is not None:
            done_out[:] = done
        for j, info in enumerate(self.infos):  # rl/rollout.py:753, 818-868: time of the landed state, episode stats
            self._time[j] = info.get("time", 0)
            self._len[j] = info.get("ep_length", 0)
            self._score[j] = info.get("ep_score", 0.0)
        return self.obs, rew, done

    @property
    def last_episode_stats(self):
        return self._time, self._len, self._score

    def landed_time(self, done):
        """Env time of the state each env is now in: what its info dict says (rl/rollout.py:753)."""
        return self._time


class HybridAsyncVectorEnv:
    def __init__(self, env_fns, max_cpus=8, verbose=False, copy=True, allow_threaded=True, context="spawn"):
        assert len(env_fns) % max_cpus == 0, \
            "Number of environments ({}) must be a multiple of the CPU count ({}).".format(len(env_fns), max_cpus)
        self.num_envs = len(env_fns)
        self.n_parallel = max_cpus
        self.n_sequential = self.num_envs // max_cpus
        self.copy = copy
        self.closed = False
        probe = env_fns[0]()  # spaces come from one throw-away instance, as gym's AsyncVectorEnv does
        space = probe.observation_space
        self.observation_space, self.action_space = space, getattr(probe, "action_space", None)
        self.obs_shape, self.obs_dtype = tuple(space.shape), np.dtype(space.dtype)
        probe.close()
        if verbose:
            print("Creating {} cpu workers with {} environments each.".format(self.n_parallel, self.n_sequential))
        ctx = mp.get_context(context)
        A = self.num_envs
        obs_bytes = int(np.prod(self.obs_shape)) * self.obs_dtype.itemsize
        self._raws = (ctx.RawArray("b", A * obs_bytes), ctx.RawArray("f", A), ctx.RawArray("B", A))
        self.obs = _as_array(self._raws[0], self.obs_dtype, (A, *self.obs_shape))  # workers write here
        self._rew = _as_array(self._raws[1], np.float32, (A,))
        self._done = _as_array(self._raws[2], np.uint8, (A,))
        self._pinned = False
        self._n_groups = 1
        self.parent_pipes, self.processes = [], []
        for i in range(self.n_parallel):
            parent, child = ctx.Pipe()
            fns = env_fns[i * self.n_sequential:(i + 1) * self.n_sequential]
            p = ctx.Process(target=_worker, name=f"ppo-env-worker-{i}", daemon=True,
                            args=(i, fns, child, self._raws, self.obs_shape, self.obs_dtype, i * self.n_sequential, A,
                                  allow_threaded))
            p.start()
            child.close()
            self.parent_pipes.append(parent)
            self.processes.append(p)

    # ------------------------------------------------------------------ plumbing
    def _gather(self, what):
        results = []
        for i, pipe in enumerate(self.parent_pipes):
            try:
                payload, ok = pipe.recv()
            except EOFError:
                raise RuntimeError(f"env worker {i} died during `{what}`") from None
            if not ok:
                self.close(terminate=True)
                raise RuntimeError(f"env worker {i} failed during `{what}`:\n{payload}")
            results.append(payload)
        return results

    def _broadcast(self, command, per_worker=None):
        for i, pipe in enumerate(self.parent_pipes):
            pipe.send((command, None if per_worker is None else per_worker[i]))
        return self._gather(command)

    def pin(self):
        """Register the shared observation block as pinned host memory (async H2D out of it).  Returns
        whether it is pinned; a refusal by the runtime only costs the overlap, never correctness."""
        if not self._pinned:
            import torch
            if torch.cuda.is_available():
                rc = torch.cuda.cudart().cudaHostRegister(self.obs.ctypes.data, self.obs.nbytes, 0)
                self._pinned = int(rc) == 0
        return self._pinned

    @property
    def obs_t(self):
        import torch
        return torch.from_numpy(self.obs)

    # ------------------------------------------------------------------ the vector-env API
    def reset(self):
        self._broadcast("reset")
        return self.obs.copy() if self.copy else self.obs

    def step(self, actions):
        actions = np.asarray(actions)
        if len(actions) != self.num_envs:
            raise ValueError(f"expected {self.num_envs} actions, got {len(actions)}")
        per = [list(actions[i * self.n_sequential:(i + 1) * self.n_sequential]) for i in range(self.n_parallel)]
        infos = [info for worker_infos in self._broadcast("step", per) for info in worker_infos]
        return (self.obs.copy() if self.copy else self.obs, self._rew.copy(), self._done.astype(bool), infos)

    # ------------------------------------------------------------------ worker groups (pipelined rollout)
    def step_async(self, group, actions):
        """Send `actions` (one per env of the group) to the workers of `group`; returns at once."""
        first_w, n_w = self._group_workers(group)
        actions = np.asarray(actions)
        if len(actions) != n_w * self.n_sequential:
            raise ValueError(f"group {group} has {n_w * self.n_sequential} envs, got {len(actions)} actions")
        for w in range(n_w):
            self.parent_pipes[first_w + w].send(("step", list(actions[w * self.n_sequential:(w + 1) * self.n_sequential])))

    def step_wait(self, group):
        """Collect the step sent by step_async: observations / rewards / dones are already in the shared block (the
        workers wrote them in place); returns the per-env info dicts of the group."""
        first_w, n_w = self._group_workers(group)
        infos = []
        for w in range(first_w, first_w + n_w):
            try:
                payload, ok = self.parent_pipes[w].recv()
            except EOFError:
                raise RuntimeError(f"env worker {w} died during `step`") from None
            if not ok:
                self.close(terminate=True)
                raise RuntimeError(f"env worker {w} failed during `step`:\n{payload}")
            infos.extend(payload)
        return infos

    def _group_workers(self, group):
        per = self.n_parallel // self._n_groups
        return group * per, per

    def groups(self, count=2):
        """The pool as `count` independently steppable groups of whole workers (contiguous env ranges), each with the
        array-stepping interface the Runner's pipelined rollout takes (`step_arrays`, `obs_t`, `last_episode_stats`):
        the GPU runs one group's policy step while the other group's workers step their envs, and the per-step upload
        is an async copy out of the group's slice of the pinned block.  [self] when the workers do not divide."""
        if count < 2 or self.n_parallel % count:
            count = 1
        self._n_groups = count
        return [PoolGroup(self, g) for g in range(count)]

    @property
    def parts(self):
        if not hasattr(self, "_parts"):
            self._parts = self.groups(2)
        return self._parts

    def seed(self, seeds=None):
        seeds = np.reshape(seeds, [self.n_parallel, self.n_sequential])
        self._broadcast("seed", [[int(s) for s in row] for row in seeds])

    def save_state(self, buffer):
        counter = 0
        for result in self._broadcast("save"):
            for _k, v in sorted(result.items()):
                buffer[f"vec_{counter:03d}"] = v
                counter += 1

    def restore_state(self, buffer):
        splits = [{f"vec_{j:03d}": buffer[f"vec_{i * self.n_sequential + j:03d}"] for j in range(self.n_sequential)}
                  for i in range(self.n_parallel)]
        self._broadcast("load", splits)

    def restore_env_state(self, env_index: int, buffer: dict):
        """Restore the envs of ONE worker from `buffer` (keys vec_000.. relative to that worker)."""
        self.parent_pipes[env_index].send(("load", buffer))
        payload, ok = self.parent_pipes[env_index].recv()
        if not ok:
            raise RuntimeError(f"env worker {env_index} failed during `load`:\n{payload}")

    def close(self, terminate=False):
        if self.closed:
            return
        self.closed = True
        if self._pinned:
            import torch
            torch.cuda.cudart().cudaHostUnregister(self.obs.ctypes.data)
            self._pinned = False
        for pipe, p in zip(self.parent_pipes, self.processes):
            try:
                if not terminate and p.is_alive():
                    pipe.send(("close", None))
                    if pipe.poll(2.0):
                        pipe.recv()
            except (BrokenPipeError, EOFError, OSError):
                pass
        for p in self.processes:
            p.join(timeout=2.0)
            if p.is_alive():
                p.terminate()  # exact process we started
                p.join(timeout=2.0)
        for pipe in self.parent_pipes:
            pipe.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
