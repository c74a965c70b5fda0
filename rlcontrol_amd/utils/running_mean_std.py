"""RunningMeanStd (mirrors utils/running_mean_std.py:2-35).

Kept for API compatibility: BaseAgent.update calls ``input_norm.update`` every step
(agents/base_agent.py:61-62).  In the reference it never influences the network (quirk Q6): the
manager constructs it as ``RunningMeanStd(state_dim)`` so ``state_dim`` lands in ``epsilon`` and
mean/var stay the scalars 0.0/1.0 that were baked into the TF graph at build time
(base_network_manager.py:37, hydra_ddpg_network.py:86-87).  The HIP kernels therefore implement
``clip((x-0)/1, state_min, state_max)`` only.
"""
import numpy as np


class RunningMeanStd(object):
    def __init__(self, epsilon=1e-4, shape=()):
        self.mean = np.zeros(shape, "float64")
        self.var = np.ones(shape, "float64")
        self.count = epsilon

    def update(self, x):
        self.update_from_moments(np.mean(x, axis=0), np.var(x, axis=0), x.shape[0])

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        total = self.count + batch_count
        delta = batch_mean - self.mean
        m2 = (self.var * self.count + batch_var * batch_count
              + np.square(delta) * self.count * batch_count / total)
        self.mean = self.mean + delta * batch_count / total
        self.var = m2 / total
        self.count = total

    def normalize(self, x):
        return (x - self.mean) / np.sqrt(self.var)

    def denormalize(self, x):
        return x * np.sqrt(self.var) + self.mean
