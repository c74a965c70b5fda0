"""Distinct uniform index sampling for the replay (mirrors utils/custom_collections.py:103-131).

The reference's ``RandomAccessQueue`` is a two-list FIFO whose ``sample(k)`` draws ``k`` DISTINCT
logical positions (0 = oldest) with ``sample_n_k`` and indexes the lists one by one.  Here the
transitions live in a device-resident SoA ring (rlcontrol_amd/csrc/replay_kernels.hip), so only the
index draw stays on the host -- and only in ``sampler="reference"`` mode, where it must reproduce
the reference's ``np.random.RandomState(seed)`` stream call for call (golden:
tests/golden/sample_n_k.json).  ``sampler="device"`` uses the Philox sampler kernel instead.
"""
import numpy as np


class DistinctIndexSampler(object):
    """``sample_n_k(n, k)``: k distinct uniform draws from range(n), reference RNG stream."""

    def __init__(self, seed=None):
        self.rng = np.random.RandomState(seed)

    def sample_n_k(self, n, k):
        if not 0 <= k <= n:
            raise ValueError("Sample larger than population or is negative")
        if k == 0:
            return np.empty((0,), dtype=np.int64)
        if 3 * k >= n:
            # dense regime: a permutation prefix (custom_collections.py:114-115)
            return self.rng.choice(n, k, replace=False)
        # sparse regime: 2k draws with replacement; the first k are the candidates and the second k
        # are spares consumed left to right whenever a candidate repeats an earlier pick; spares are
        # redrawn when exhausted (custom_collections.py:116-131)
        draws = self.rng.choice(n, 2 * k)
        seen = set()
        spare = k
        for pos in range(k):
            cand = draws[pos]
            while cand in seen:
                cand = draws[pos] = draws[spare]
                spare += 1
                if spare == 2 * k:
                    draws[k:] = self.rng.choice(n, k)
                    spare = k
            seen.add(cand)
        return draws[:k]
