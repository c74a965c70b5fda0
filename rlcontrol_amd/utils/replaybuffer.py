"""Replay buffer front-end (mirrors utils/replaybuffer.py:11-42 of the reference).

Same constructor and methods as the reference's ``ReplayBuffer`` -- ``add``, ``get_size``,
``sample_batch`` -- but the transitions are stored in the device-resident SoA ring owned by the
agent's HIP handle (rlcontrol_amd/csrc/replay_kernels.hip); nothing is kept on the host except the
index sampler.  ``sample_batch`` returns the reference's five float64 arrays
``(state[B,S], action[B,A], reward[B], next_state[B,S], gamma[B])`` gathered on the GPU;
the fused learn path (BaseAgent.learn) only needs ``sample_indices``.
"""
from collections import namedtuple

import numpy as np

from .custom_collections import DistinctIndexSampler

Transition = namedtuple('Transition', ['state', 'action', 'reward', 'next_state', 'transition_gamma'])


class ReplayBuffer(object):
    def __init__(self, buffer_size, random_seed, store=None, sampler="reference"):
        """store: (DDPGPopulation-like handle, agent index) that owns the device ring."""
        if store is None:
            raise RuntimeError("ReplayBuffer needs the device store of a HIP agent handle; "
                               "rlcontrol_amd keeps no host-side transition storage")
        self.buffer_size = int(buffer_size)
        self._pop, self._agent = store
        if sampler not in ("reference", "device"):
            raise ValueError("replay sampler must be 'reference' or 'device'")
        self.sampler_mode = sampler
        # same stream as RandomAccessQueue(maxlen, seed=random_seed).rng (custom_collections.py:15)
        self.sampler = DistinctIndexSampler(random_seed)

    def add(self, state, action, reward, next_state, transition_gamma):
        self._pop.replay_add(self._agent, state, action, reward, next_state, transition_gamma)

    def get_size(self):
        return self._pop.replay_size(self._agent)

    def sample_indices(self, batch_size):
        """k distinct logical positions (0 = oldest), reference RNG stream or device Philox."""
        n = self.get_size()
        assert n >= batch_size
        if self.sampler_mode == "device":
            return self._pop.replay_sample_indices(self._agent, batch_size)
        return np.asarray(self.sampler.sample_n_k(n, batch_size), dtype=np.int64)

    def sample_batch(self, batch_size):
        idx = self.sample_indices(batch_size)
        s, a, r, s2, g = self._pop.replay_gather(self._agent, idx)
        return s, a, r, s2, g

    def clear(self):
        raise NotImplementedError("the reference's clear() is unused (utils/replaybuffer.py:39-42)")
