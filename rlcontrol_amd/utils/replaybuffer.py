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
        self._pre = None          # (RNG state before the draw, size it assumed, batch size, indices): presample()

    def add(self, state, action, reward, next_state, transition_gamma):
        self._pop.replay_add(self._agent, state, action, reward, next_state, transition_gamma)

    def get_size(self):
        return self._pop.replay_size(self._agent)

    def presample(self, batch_size, n_expected):
        """Draw the indices of the NEXT sample_indices(batch_size) call now -- while the GPU works on the update just
        launched -- assuming the buffer will hold n_expected transitions then.  The sampler's stream is the only user of
        its RandomState, so drawing early changes nothing; if the guess turns out wrong (a truncated step stores
        nothing, utils/replaybuffer.py / agents/base_agent.py:56-58) the draw is undone and made again."""
        if self.sampler_mode != "reference" or n_expected < batch_size:
            return
        state = self.sampler.rng.get_state()
        self._pre = (state, int(n_expected), int(batch_size),
                     np.asarray(self.sampler.sample_n_k(int(n_expected), batch_size), dtype=np.int64))

    def sample_indices(self, batch_size):
        """k distinct logical positions (0 = oldest), reference RNG stream or device Philox."""
        n = self.get_size()
        assert n >= batch_size
        if self.sampler_mode == "device":
            return self._pop.replay_sample_indices(self._agent, batch_size)
        pre, self._pre = self._pre, None
        if pre is not None:
            state, n_expected, k, idx = pre
            if n_expected == n and k == batch_size:
                return idx
            self.sampler.rng.set_state(state)          # wrong guess: the stream goes back to where it was
        return np.asarray(self.sampler.sample_n_k(n, batch_size), dtype=np.int64)

    def sample_batch(self, batch_size):
        idx = self.sample_indices(batch_size)
        s, a, r, s2, g = self._pop.replay_gather(self._agent, idx)
        return s, a, r, s2, g

    def clear(self):
        raise NotImplementedError("the reference's clear() is unused (utils/replaybuffer.py:39-42)")
