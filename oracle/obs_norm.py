"""ORACLE (test infrastructure — never imported by the product): CPU restatement of the reference's
observation normalisation.

  * RunningMeanStd                rl/utils.py:379-455 (update_mean_var_count_from_moments + the class)
  * prep                          rl/models.py:824-856, observation_scaling "scaled": uint8 -> float32 x / 255
  * batch_moments                 rl/models.py:681-685: float32 mean / biased variance over the batch axis
  * ObsNormalizer.update / apply  rl/models.py:661-694: update the running statistics from one batch, refresh
                                  mu = float32(mean), std = float32(var) ** 0.5, and
                                  clamp((x - mu) / (std + eps), -5, 5)

Pinned by tests/golden/obsnorm_golden.npz (the reference's TVFModel.perform_normalization run in the build
container, tests/golden/make_obsnorm_golden.py).  The reference reduces a batch with torch's float32 mean /
var; this restatement reduces in float64 and rounds to float32, which differs from torch's summation order by
at most a few float32 ulps of the batch mean — the golden comparison carries that tolerance (1e-6).
"""
import numpy as np


class RunningMeanStd:
    def __init__(self, epsilon=1e-4, shape=()):
        self.mean = np.zeros(shape, np.float64)
        self.var = np.ones(shape, np.float64)
        self.count = epsilon

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        delta = batch_mean - self.mean
        tot = self.count + batch_count
        new_mean = self.mean + delta * batch_count / tot
        m2 = self.var * self.count + batch_var * batch_count + np.square(delta) * self.count * batch_count / tot
        self.mean, self.var, self.count = new_mean, m2 / tot, tot


def prep(x):
    x = np.asarray(x)
    if x.dtype == np.uint8:
        return x.astype(np.float32) / np.float32(255.0)
    return x.astype(np.float32, copy=False)


def batch_moments(xp):
    x64 = xp.astype(np.float64)
    return x64.mean(axis=0).astype(np.float32), x64.var(axis=0).astype(np.float32)


class ObsNormalizer:
    def __init__(self, input_dims, norm_eps=1e-5):
        self.rms = RunningMeanStd(shape=tuple(input_dims))
        self.norm_eps = np.float32(norm_eps)
        self.refresh()

    def refresh(self):
        self.mu = self.rms.mean.astype(np.float32)
        self.std = self.rms.var.astype(np.float32) ** np.float32(0.5)

    def update(self, x):
        xp = prep(x)
        bm, bv = batch_moments(xp)
        self.rms.update_from_moments(bm, bv, xp.shape[0])
        self.refresh()

    def apply(self, x):
        xp = prep(x)
        return np.clip((xp - self.mu) / (self.std + self.norm_eps), np.float32(-5), np.float32(5)).astype(np.float32)
