"""Reference-structured CPU path (test infrastructure; bench.py's "cpu_baseline" leg and the
agent-level parity tests).

Keeps the reference's structure on the host side:
  * list-of-records replay with FIFO eviction, ``sample_n_k`` distinct sampling and the 5-array
    conversion per minibatch (utils/replaybuffer.py:25-37, utils/custom_collections.py:83-131);
  * BaseAgent's insert rule / learn gate (agents/base_agent.py:54-70);
  * OU exploration on the greedy action (agents/DDPG.py:36-44, utils/exploration_policy.py:18-21);
  * float64 TD glue + the seven-step update, here one call into the C restatement
    (oracle/ddpg_oracle.c) where the reference makes seven TF-1.15 Session.run calls.
TensorFlow-1.15 itself cannot run here or on the GPU box, so this "port" is
what gets timed next to the MI355X number; it is a baseline, not a target.
"""
import collections

import numpy as np

from .ddpg import DDPGOracle, Dims, init_params

Transition = collections.namedtuple('Transition', ['state', 'action', 'reward', 'next_state', 'transition_gamma'])


class ListReplay(object):
    """FIFO of records + the reference's index sampler (restated; pinned by tests/golden/sample_n_k.json)."""

    def __init__(self, maxlen, seed):
        self.maxlen = int(maxlen)
        self.items = collections.deque()
        self.rng = np.random.RandomState(seed)

    def __len__(self):
        return len(self.items)

    def append(self, x):
        self.items.append(x)
        if len(self.items) > self.maxlen:
            self.items.popleft()

    def sample_n_k(self, n, k):
        if not 0 <= k <= n:
            raise ValueError("Sample larger than population or is negative")
        if k == 0:
            return np.empty((0,), dtype=np.int64)
        if 3 * k >= n:
            return self.rng.choice(n, k, replace=False)
        result = self.rng.choice(n, 2 * k)
        selected = set()
        j = k
        for i in range(k):
            x = result[i]
            while x in selected:
                x = result[i] = result[j]
                j += 1
                if j == 2 * k:
                    result[k:] = self.rng.choice(n, k)
                    j = k
            selected.add(x)
        return result[:k]

    def sample_batch(self, k):
        idx = self.sample_n_k(len(self.items), k)
        batch = [self.items[int(i)] for i in idx]
        return tuple(map(np.array, zip(*batch))), idx


class CpuDDPGAgent(object):
    """start/step/update/reset agent whose arithmetic is the C oracle."""

    def __init__(self, S, A, H1, HA, HC, batch_size, buffer_size, gamma, tau, actor_lr, critic_lr, state_min,
                 state_max, action_min, action_max, seed, warmup_steps=0, ou=(0.15, 0.0, 0.2), clip_state=True,
                 theta0=None):
        self.d = Dims(S, A, H1, HA, HC)
        th = init_params(self.d, seed) if theta0 is None else np.asarray(theta0, np.float32)
        self.net = DDPGOracle(self.d, th, actor_lr, critic_lr, tau, state_min, state_max, action_max, clip_state)
        self.replay = ListReplay(buffer_size, seed)
        self.batch_size, self.warmup_steps, self.gamma = batch_size, warmup_steps, gamma
        self.amin, self.amax = np.asarray(action_min, np.float64), np.asarray(action_max, np.float64)
        self.ou_theta, self.ou_mu, self.ou_sigma = ou
        self.ou_rng = np.random.RandomState(seed)
        self.noise = self.ou_mu
        self.n_updates = 0
        self.last_idx = None

    def reset(self):
        self.noise = self.ou_mu

    def _act(self, state, is_train):
        greedy = self.net.act(np.asarray(state, np.float64)[None, :])[0]
        if not is_train:
            return greedy
        draw = self.ou_rng.normal(self.ou_mu * np.ones(self.d.A), self.ou_sigma * np.ones(self.d.A))
        self.noise = self.noise + (draw - self.noise * self.ou_theta)
        return np.clip(greedy + self.noise, self.amin, self.amax)

    def start(self, state, is_train):
        return self._act(state, is_train)

    def step(self, state, is_train):
        return self._act(state, is_train)

    def update(self, state, next_state, reward, action, is_terminal, is_truncated):
        if not is_truncated:
            self.replay.append(Transition(state, action, reward, next_state, 0.0 if is_terminal else self.gamma))
        if len(self.replay) > max(self.warmup_steps, self.batch_size):
            (s, a, r, s2, g), idx = self.replay.sample_batch(self.batch_size)
            self.last_idx = idx
            self.net.update(s, a, s2, r, g)
            self.n_updates += 1


def synthetic_pendulum_replay(n, seed=0):
    """BASELINE.md section 3 / SURVEY.md 8(d): n Pendulum-shaped transitions from RandomState(seed)."""
    rng = np.random.RandomState(seed)
    th = rng.uniform(-np.pi, np.pi, n)
    thd = rng.uniform(-8.0, 8.0, n)
    a = rng.uniform(-2.0, 2.0, n)
    s = np.stack([np.cos(th), np.sin(th), thd], 1)
    r = -(th ** 2 + 0.1 * thd ** 2 + 0.001 * a ** 2)
    thd2 = thd + (-3 * 10.0 / 2 * np.sin(th + np.pi) + 3.0 * a) * 0.05
    th2 = th + thd2 * 0.05
    thd2 = np.clip(thd2, -8.0, 8.0)
    s2 = np.stack([np.cos(th2), np.sin(th2), thd2], 1)
    g = np.full(n, 0.99)
    return s, a[:, None], r, s2, g
