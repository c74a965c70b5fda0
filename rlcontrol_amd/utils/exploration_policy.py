"""Host-side exploration noise (mirrors utils/exploration_policy.py:4-73).

``OrnsteinUhlenbeckProcess`` reproduces the reference's RandomState stream draw for draw
(``n += N(mu, sigma) - theta*n`` then ``clip(greedy + n)``; reset -> n = mu), which is what the
drop-in single-agent path uses so that trajectories are comparable with the reference's.  The
device-resident generator used by the vectorised path is the HIP kernel ``rlc_ou_noise_kernel``
(rlcontrol_amd/csrc/act_kernels.hip): same recurrence, Philox normals, statistical parity only.
"""
import numpy as np


class OrnsteinUhlenbeckProcess(object):
    def __init__(self, random_seed, action_dim, action_min, action_max, theta, mu, sigma):
        self.rng = np.random.RandomState(random_seed)
        self.action_dim = action_dim
        self.action_min = action_min
        self.action_max = action_max
        self.theta = theta
        self.mu = mu
        self.sigma = sigma
        self.noise_t = self.mu

    def generate(self, greedy_action, step):
        draw = self.rng.normal(self.mu * np.ones(self.action_dim), self.sigma * np.ones(self.action_dim))
        self.noise_t = self.noise_t + (draw - self.noise_t * self.theta)
        return np.clip(greedy_action + self.noise_t, self.action_min, self.action_max)

    def reset(self):
        self.noise_t = self.mu


class RandomUniform(object):
    def __init__(self, random_seed, action_min, action_max, is_continuous):
        self.rng = np.random.RandomState(random_seed)
        self.action_min = action_min
        self.action_max = action_max
        self.is_continuous = is_continuous

    def generate(self, greedy_action, step):
        if self.is_continuous:
            return self.rng.uniform(self.action_min, self.action_max)
        return self.rng.choice(range(int(self.action_max - self.action_min + 1)))

    def reset(self):
        pass


class EpsilonGreedy(object):
    def __init__(self, random_seed, action_min, action_max, annealing_steps, min_epsilon, max_epsilon,
                 is_continuous):
        self.rng = np.random.RandomState(random_seed)
        self.action_min = action_min
        self.action_max = action_max
        self.epsilon = max_epsilon
        self.min_epsilon = min_epsilon
        self.annealing_steps = annealing_steps
        self.epsilon_step = -(self.epsilon - self.min_epsilon) / float(self.annealing_steps)
        self.is_continuous = is_continuous

    def generate(self, greedy_action, step):
        epsilon = max(self.min_epsilon, self.epsilon_step * step + self.epsilon)
        if self.rng.random_sample() < epsilon:
            if self.is_continuous:
                return self.rng.uniform(self.action_min, self.action_max)
            return self.rng.choice(range(int(self.action_max - self.action_min + 1)))
        return greedy_action

    def reset(self):
        pass
