"""Host-side exploration policies behind the reference's names (utils/exploration_policy.py of the reference):
``OrnsteinUhlenbeckProcess``, ``EpsilonGreedy``, ``RandomUniform``, each with ``generate(greedy_action, step)``
and ``reset()``, plus ``make_policy`` which maps ``config.exploration_policy`` to an instance.

What is pinned (tests/golden/ou_noise.json, generated from the reference): the OU process consumes one
``RandomState.normal(mu * 1, sigma * 1)`` vector per call and forms ``noise + (draw - theta * noise)`` before
clipping ``greedy + noise`` -- draw for draw and bit for bit the reference stream, which is what makes the
drop-in single-agent path comparable with the reference's trajectories.  The device-resident generator of the
vectorised path (``ddpg_ou_explore`` in rlcontrol_amd/csrc/ddpg_policy.h) uses the same recurrence on Philox
normals: statistical parity only.
"""
import numpy as np


class _SeededPolicy(object):
    """Common part: a private MT19937 stream and the action box."""

    def __init__(self, random_seed, action_min, action_max):
        self.rng = np.random.RandomState(random_seed)
        self.action_min, self.action_max = action_min, action_max

    def _any_action(self, continuous):
        lo, hi = self.action_min, self.action_max
        if continuous:
            return self.rng.uniform(lo, hi)
        return self.rng.choice(range(int(hi - lo + 1)))

    def reset(self):
        """stateless policies have nothing to forget between episodes"""


class OrnsteinUhlenbeckProcess(_SeededPolicy):
    """Temporally correlated noise added to the greedy action (DDPG's default exploration)."""

    def __init__(self, random_seed, action_dim, action_min, action_max, theta, mu, sigma):
        _SeededPolicy.__init__(self, random_seed, action_min, action_max)
        self.action_dim = action_dim
        self.theta, self.mu, self.sigma = theta, mu, sigma
        self._loc = mu * np.ones(action_dim)
        self._scale = sigma * np.ones(action_dim)
        self.noise_t = mu

    def generate(self, greedy_action, step):
        kick = self.rng.normal(self._loc, self._scale)
        pulled_back = self.noise_t * self.theta
        self.noise_t = self.noise_t + (kick - pulled_back)
        return np.clip(greedy_action + self.noise_t, self.action_min, self.action_max)

    def reset(self):
        self.noise_t = self.mu


class RandomUniform(_SeededPolicy):
    """Ignores the greedy action altogether."""

    def __init__(self, random_seed, action_min, action_max, is_continuous):
        _SeededPolicy.__init__(self, random_seed, action_min, action_max)
        self.is_continuous = is_continuous

    def generate(self, greedy_action, step):
        return self._any_action(self.is_continuous)


class EpsilonGreedy(_SeededPolicy):
    """Random action with a probability annealed linearly from max_epsilon to min_epsilon over annealing_steps."""

    def __init__(self, random_seed, action_min, action_max, annealing_steps, min_epsilon, max_epsilon,
                 is_continuous):
        _SeededPolicy.__init__(self, random_seed, action_min, action_max)
        self.is_continuous = is_continuous
        self.min_epsilon, self.epsilon = min_epsilon, max_epsilon
        self.annealing_steps = annealing_steps
        self.epsilon_step = -(max_epsilon - min_epsilon) / float(annealing_steps)

    def current_epsilon(self, step):
        return max(self.min_epsilon, self.epsilon + self.epsilon_step * step)

    def generate(self, greedy_action, step):
        explore = self.rng.random_sample() < self.current_epsilon(step)
        return self._any_action(self.is_continuous) if explore else greedy_action


def make_policy(config, random_seed, action_dim, action_min, action_max):
    """``config.exploration_policy`` -> (uses_external_exploration, policy or None); ValueError for unknown names
    (agents/network/base_network_manager.py:43-73 of the reference)."""
    kind = config.exploration_policy
    if kind == 'none':
        return False, None
    if kind == 'ou_noise':
        return True, OrnsteinUhlenbeckProcess(random_seed, action_dim, action_min, action_max, theta=config.ou_theta,
                                              mu=config.ou_mu, sigma=config.ou_sigma)
    if kind == 'epsilon_greedy':
        return True, EpsilonGreedy(random_seed, action_min, action_max, config.annealing_steps, config.min_epsilon,
                                   config.max_epsilon, is_continuous=True)
    if kind == 'random_uniform':
        return True, RandomUniform(random_seed, action_min, action_max, is_continuous=True)
    raise ValueError("Invalid Value for config.exploration_policy")
