"""GPU parity of the device replay ring / gather / sampler (through the C ABI).
Bit-exact: this is copy and index work."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pop(n_agents=1, cap=8, B=4, S=3, A=1, seeds=None):
    from rlcontrol_amd.hip_ddpg import DDPGPopulation
    return DDPGPopulation(n_agents, S, A, 16, 16, 16, B, cap, 0.01, -np.ones(S) * 8, np.ones(S) * 8,
                          -2 * np.ones(A), 2 * np.ones(A), 1e-3, 1e-2,
                          seeds=seeds if seeds is not None else list(range(1, n_agents + 1)))


def test_fifo_eviction_and_gather_match_reference_roundtrip(hip_lib, golden_dir):
    """capacity-8 ring, 21 adds: logical order after every add and the sampled batches equal what the
    reference's ReplayBuffer returned (tests/golden/replay_roundtrip.json)."""
    from rlcontrol_amd.utils.replaybuffer import ReplayBuffer
    with open(os.path.join(golden_dir, "replay_roundtrip.json")) as f:
        log = json.load(f)
    pop = _pop(cap=8, B=4)
    rb = ReplayBuffer(8, 3, store=(pop, 0))
    for e in log:
        s, a, r, s2, g = e["add"]
        rb.add(np.array(s), np.array(a), r, np.array(s2), g)
        assert rb.get_size() == e["size"]
        # logical content oldest -> newest
        _, _, rr, _, _ = pop.replay_gather(0, np.arange(e["size"]))
        want = np.array(e["content_reward"])
        assert np.array_equal(rr, want)          # rewards are stored as float64: bit-exact
        if "sample" in e:
            st, ac, rw, ns, gm = rb.sample_batch(4)
            smp = e["sample"]
            assert [list(x.shape) for x in (st, ac, rw, ns, gm)] == smp["shapes"]
            assert [str(x.dtype) for x in (st, ac, rw, ns, gm)] == smp["dtypes"]
            assert np.array_equal(rw, np.array(smp["reward"]))
            assert np.array_equal(gm, np.array(smp["gamma"]))
            # states/actions pass through the fp32 placeholder cast
            assert np.array_equal(st, np.array(smp["state"]).astype(np.float32).astype(np.float64))
            assert np.array_equal(ns, np.array(smp["next_state"]).astype(np.float32).astype(np.float64))
            assert np.array_equal(ac, np.array(smp["action"]).astype(np.float32).astype(np.float64))
    pop.close()


def test_add_batch_equals_repeated_add_including_overflow(hip_lib):
    rng = np.random.RandomState(0)
    n = 37
    s, s2 = rng.randn(n, 3), rng.randn(n, 3)
    a, r, g = rng.randn(n, 1), rng.randn(n), rng.rand(n)
    p1, p2 = _pop(cap=10), _pop(cap=10)
    for i in range(n):
        p1.replay_add(0, s[i], a[i], r[i], s2[i], g[i])
    p2.replay_add_batch(0, s[:5], a[:5], r[:5], s2[:5], g[:5])
    p2.replay_add_batch(0, s[5:30], a[5:30], r[5:30], s2[5:30], g[5:30])    # longer than the capacity
    p2.replay_add_batch(0, s[30:], a[30:], r[30:], s2[30:], g[30:])
    assert p1.replay_size(0) == p2.replay_size(0) == 10
    g1 = p1.replay_gather(0, np.arange(10))
    g2 = p2.replay_gather(0, np.arange(10))
    for x, y in zip(g1, g2):
        assert np.array_equal(x, y)
    assert np.array_equal(g1[2], r[-10:])
    p1.close(); p2.close()


def test_gather_errors_and_empty(hip_lib):
    from rlcontrol_amd._lib import RlcError
    pop = _pop(cap=8)
    out = pop.replay_gather(0, np.zeros(0, np.int64))
    assert out[0].shape == (0, 3)
    with pytest.raises(RlcError, match="index out of range"):
        pop.replay_gather(0, np.array([0]))
    pop.replay_add(0, np.zeros(3), np.zeros(1), 1.0, np.ones(3), 0.99)
    with pytest.raises(RlcError, match="index out of range"):
        pop.replay_gather(0, np.array([1]))
    with pytest.raises(RlcError, match="Sample larger than population"):
        pop.replay_sample_indices(0, 2)
    with pytest.raises(RlcError, match="replay holds"):
        pop.update(1)
    pop.close()


def test_full_size_ring_gather_is_exact(hip_lib):
    """BASELINE size: 1e6 transitions per agent, two agents; gather of random logical indices is bit-exact
    against a host mirror, before and after the ring wraps."""
    N = 10 ** 6
    rng = np.random.RandomState(5)
    s = rng.randn(N, 3).astype(np.float32).astype(np.float64)
    s2 = rng.randn(N, 3).astype(np.float32).astype(np.float64)
    a = rng.randn(N, 1).astype(np.float32).astype(np.float64)
    r, g = rng.randn(N), rng.rand(N)
    pop = _pop(n_agents=2, cap=N, B=100)
    for ag in range(2):
        pop.replay_add_batch(ag, s, a, r + ag, s2, g)
    idx = rng.choice(N, 4096, replace=False)
    for ag in range(2):
        gs, ga, gr, gs2, gg = pop.replay_gather(ag, idx)
        assert np.array_equal(gs, s[idx]) and np.array_equal(gs2, s2[idx]) and np.array_equal(ga, a[idx])
        assert np.array_equal(gr, r[idx] + ag) and np.array_equal(gg, g[idx])
    # wrap: 1000 more adds evict the 1000 oldest
    pop.replay_add_batch(0, s[:1000] + 1, a[:1000], r[:1000] - 7, s2[:1000], g[:1000])
    assert pop.replay_size(0) == N
    _, _, gr, _, _ = pop.replay_gather(0, np.array([0, N - 1001, N - 1000, N - 1]))
    assert np.array_equal(gr, np.array([r[1000], r[N - 1], r[0] - 7, r[999] - 7]))
    pop.close()


def test_device_sampler_distinct_uniform(hip_lib):
    """sample_n_k on the device (Philox): k distinct in-range indices; chi-square uniformity; both regimes."""
    pop = _pop(cap=5000, B=100, seeds=[77])
    rng = np.random.RandomState(0)
    n = 3000
    pop.replay_add_batch(0, rng.randn(n, 3), rng.randn(n, 1), rng.randn(n), rng.randn(n, 3), rng.rand(n))
    counts = np.zeros(n)
    draws = 300
    seen = set()
    for _ in range(draws):
        idx = pop.replay_sample_indices(0, 100)
        assert idx.min() >= 0 and idx.max() < n and len(set(idx.tolist())) == 100
        counts[idx] += 1
        seen.add(tuple(idx.tolist()))
    assert len(seen) == draws                      # the call counter advances the stream
    expected = draws * 100 / n
    chi2 = ((counts - expected) ** 2 / expected).sum()
    assert abs(chi2 - n) < 6 * np.sqrt(2 * n)      # chi2 ~ N(n, 2n)
    pop.close()
    # dense regime (3k >= n): permutation prefix
    pop = _pop(cap=128, B=100, seeds=[5])
    pop.replay_add_batch(0, rng.randn(101, 3), rng.randn(101, 1), rng.randn(101), rng.randn(101, 3), rng.rand(101))
    hits = np.zeros(101)
    for _ in range(200):
        idx = pop.replay_sample_indices(0, 100)
        assert len(set(idx.tolist())) == 100 and idx.max() < 101
        hits[idx] += 1
    assert hits.min() > 170                        # each index is left out ~1/101 of the time
    pop.close()
