"""Running mean / variance in float64 (reference: rl/utils.py:379-455 `RunningMeanStd`,
`update_mean_var_count_from_moments`): Chan et al.'s pairwise merge of (mean, var, count) with a
batch's moments.  Used by the reward normaliser (ppo_amd/wrappers.py) and, under data parallelism,
fed with moments that were all-reduced over ranks first (ppo_amd/parallel.py)."""
import numpy as np


def merge_moments(mean, var, count, batch_mean, batch_var, batch_count):
    """(mean, var, count) of the union of a population and a batch, from their moments."""
    total = count + batch_count
    delta = batch_mean - mean
    merged_mean = mean + delta * batch_count / total
    m2 = var * count + batch_var * batch_count + np.square(delta) * count * batch_count / total
    return merged_mean, m2 / total, total


class RunningMeanStd:
    """mean / var are float64 arrays of `shape`; count starts at epsilon so the first batch dominates."""

    def __init__(self, epsilon=1e-4, shape=()):
        self.restore_state((np.zeros(shape, "float64"), np.ones(shape, "float64"), epsilon))

    def update(self, x):
        if type(x) in (float, int):  # a scalar is a batch of one with no spread
            self.update_from_moments(x, 0, 1)
        else:
            self.update_from_moments(np.mean(x, axis=0), np.var(x, axis=0), x.shape[0])

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        self.restore_state(merge_moments(*self.save_state(), batch_mean, batch_var, batch_count))

    def save_state(self):
        return (self.mean, self.var, self.count)

    def restore_state(self, state):
        self.mean, self.var, self.count = state
