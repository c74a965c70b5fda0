"""Pendulum-v0, restated from the public gym 0.18.0 definition (classic_control/pendulum.py).

gym is a third-party dependency of the reference (requirements.txt:9) that is absent from this
image and from /root/reference, so its dynamics are restated here --
"parity unpinned": no reference fixture holds Pendulum trajectories.  Constants: g=10, m=l=1,
dt=0.05, |torque|<=2, |speed|<=8, reward = -(wrap(th)^2 + 0.1 thdot^2 + 0.001 u^2) using the
pre-step state, reset th~U(-pi,pi), thdot~U(-1,1), TimeLimit 200 steps -> done=True.
The seeding path (gym.utils.seeding.np_random: sha512 of the decimal seed string, first 8 bytes,
split into 32-bit words, fed to RandomState.seed) is restated too so that a given seed yields the
start states gym 0.18 would give.
"""
import hashlib
import struct

import numpy as np


def _gym_np_random(seed):
    seed = int(seed) % 2 ** 64
    digest = hashlib.sha512(str(seed).encode('utf8')).digest()[:8]
    words = struct.unpack("<2I", digest)
    big = words[0] + (words[1] << 32)
    ints = []
    while big > 0:
        big, mod = divmod(big, 2 ** 32)
        ints.append(mod)
    rng = np.random.RandomState()
    rng.seed(ints if ints else [0])
    return rng


class _Box(object):
    def __init__(self, low, high):
        self.low = np.asarray(low, np.float32)
        self.high = np.asarray(high, np.float32)
        self.shape = self.low.shape
        self._rng = np.random.RandomState()

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(np.float32)


class PendulumEnv(object):
    max_speed = 8.0
    max_torque = 2.0
    dt = 0.05
    g = 10.0
    m = 1.0
    l = 1.0

    def __init__(self):
        self._max_episode_steps = 200
        self.action_space = _Box([-self.max_torque], [self.max_torque])
        self.observation_space = _Box([-1.0, -1.0, -self.max_speed], [1.0, 1.0, self.max_speed])
        self.np_random = _gym_np_random(0)
        self.state = None
        self._elapsed = 0

    def seed(self, seed=None):
        self.np_random = _gym_np_random(0 if seed is None else seed)
        return [seed]

    def _obs(self):
        th, thdot = self.state
        return np.array([np.cos(th), np.sin(th), thdot])

    def reset(self):
        high = np.array([np.pi, 1.0])
        self.state = self.np_random.uniform(low=-high, high=high)
        self._elapsed = 0
        return self._obs()

    def step(self, u):
        th, thdot = self.state
        u = np.clip(u, -self.max_torque, self.max_torque)[0]
        wrapped = ((th + np.pi) % (2 * np.pi)) - np.pi
        cost = wrapped ** 2 + 0.1 * thdot ** 2 + 0.001 * (u ** 2)
        new_thdot = thdot + (-3 * self.g / (2 * self.l) * np.sin(th + np.pi)
                             + 3.0 / (self.m * self.l ** 2) * u) * self.dt
        new_th = th + new_thdot * self.dt
        new_thdot = np.clip(new_thdot, -self.max_speed, self.max_speed)
        self.state = np.array([new_th, new_thdot])
        self._elapsed += 1
        done = self._elapsed >= self._max_episode_steps     # gym.wrappers.TimeLimit
        info = {'TimeLimit.truncated': True} if done else {}
        return self._obs(), -cost, done, info

    def close(self):
        pass
