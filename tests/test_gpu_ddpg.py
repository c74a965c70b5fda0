"""GPU parity of the fused DDPG hot path against the CPU oracle (through the C ABI).

Tolerance: north_star asks Q-values and policy-gradient directions within 1e-5 relative (fp32) of the
reference arithmetic on identical minibatches.  "rel" below is max|x-y| / max|y| over the tensor.
Parameters after an Adam step are compared more loosely where noted: Adam's m/(sqrt(v)+eps) turns a
gradient element that is ~0 up to rounding into a +-lr step of arbitrary sign (the oracle's own
fp32-vs-float64 restatements differ the same way, tests/test_oracle.py).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SMIN, SMAX, AMIN, AMAX = [-1, -1, -8], [1, 1, 8], [-2.0], [2.0]


def _rel(x, y):
    x, y = np.asarray(x, np.float64).ravel(), np.asarray(y, np.float64).ravel()
    return float(np.max(np.abs(x - y)) / (np.max(np.abs(y)) + 1e-30))


def _cos(x, y):
    x, y = np.asarray(x, np.float64).ravel(), np.asarray(y, np.float64).ravel()
    return float(x @ y / (np.linalg.norm(x) * np.linalg.norm(y) + 1e-300))


def _make(dims, B, n_agents=1, cap=4096, lr=(1e-3, 1e-2), kernel="auto", seeds=None, smin=None, smax=None,
          amax=None):
    from rlcontrol_amd.hip_ddpg import DDPGPopulation
    S, A, H1, HA, HC = dims
    smin = -np.ones(S) * 2 if smin is None else smin
    smax = np.ones(S) * 2 if smax is None else smax
    amax = np.linspace(1.0, 2.0, A) if amax is None else np.asarray(amax, np.float64)
    pop = DDPGPopulation(n_agents, S, A, H1, HA, HC, B, cap, 0.01, smin, smax, -amax, amax, lr[0], lr[1],
                         seeds=seeds if seeds is not None else list(range(10, 10 + n_agents)))
    if kernel != "auto":
        pop.set_kernel(kernel)
    return pop, smin, smax, amax


def _oracle(dims, th, lr, smin, smax, amax):
    from oracle.ddpg import DDPGOracle, Dims
    return DDPGOracle(Dims(*dims), th, lr[0], lr[1], 0.01, smin, smax, amax)


def _batch(rng, B, S, A):
    s = rng.uniform(-3, 3, (B, S))
    a = rng.uniform(-2, 2, (B, A))
    s2 = rng.uniform(-3, 3, (B, S))
    r = rng.uniform(-16, 0, B)
    g = np.where(rng.rand(B) < 0.2, 0.0, 0.99)
    return s, a, s2, r, g


KERNELS = ["generic", "mfma"]
CASES = [((3, 1, 200, 200, 200), 100), ((3, 1, 200, 200, 200), 32), ((8, 2, 200, 200, 200), 64),
         ((8, 2, 64, 48, 40), 17), ((1, 1, 16, 16, 16), 5), ((3, 1, 128, 128, 128), 128),
         # batches that end inside the first four rows of the seventh tile run the tail-of-four kernels (mfma_blocks.h T4;
         # 100 above is the full tail): a one-row and a three-row tail, and 101 = the first batch that must not take them
         ((3, 1, 200, 200, 200), 97), ((8, 2, 200, 160, 144), 99), ((8, 2, 200, 160, 144), 101)]


def _skip_unless_supported(pop, kernel):
    """select the kernel under test explicitly ('auto' would pick mfma wherever it is supported)"""
    from rlcontrol_amd._lib import RlcError
    try:
        pop.set_kernel(kernel)
    except RlcError:
        pop.close()
        pytest.skip("MFMA kernel does not cover these dimensions (generic kernel does)")
    assert pop.kernel_in_use() == kernel


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("dims,B", CASES)
def test_single_update_taps_and_gradients(hip_lib, dims, B, kernel):
    """One update_network call from identical weights / optimizer state / minibatch."""
    from oracle.ddpg import Dims, init_params
    pop, smin, smax, amax = _make(dims, B)
    _skip_unless_supported(pop, kernel)
    pop.enable_grad_taps(True)
    th = init_params(Dims(*dims), 3)
    pop.set_params(0, th)
    o = _oracle(dims, th, (1e-3, 1e-2), smin, smax, amax)
    rng = np.random.RandomState(11)
    s, a, s2, r, g = _batch(rng, B, dims[0], dims[1])
    pop.update_batch(0, s, a, s2, r, g)
    taps = o.update(s, a, s2, r, g, taps=True)
    for name in ("q", "y", "a_out", "dqda"):
        assert _rel(pop.last_tap(0, name), taps[name]) < 1e-5, name
    lay, P = Dims(*dims).layout()
    for which in ("grads_c", "grads_a"):
        got, want = pop.last_tap(0, which), taps[which]
        assert _cos(got, want) > 1 - 1e-9, which                 # policy-gradient DIRECTION
        for name, (off, shp) in lay.items():
            n = int(np.prod(shp))
            if np.any(want[off:off + n]):
                assert _rel(got[off:off + n], want[off:off + n]) < 1e-5, (which, name)
            else:
                assert not np.any(got[off:off + n]), (which, name)   # None gradients stay untouched
    # Polyak and beta powers are exact-ish functions of the above
    assert _rel(pop.get_blob(0, "theta_target"), o.theta_t) < 1e-5
    assert np.allclose(pop.get_beta_powers(0), o.pw, rtol=1e-6)
    # parameters: compare where the Adam step is well conditioned (|g| not ~ rounding noise)
    th_gpu = pop.get_blob(0, "theta")
    for which, lr in (("grads_c", 1e-2), ("grads_a", 1e-3)):
        solid = np.abs(taps[which]) > 1e-4 * np.max(np.abs(taps[which]))
        lo = lay["Wc2"][0]
        sel = solid.copy()
        if which == "grads_c":
            sel[:lo] = False      # W1/b1 get a second (actor) step on top: checked through theta as a whole
        else:
            sel[:lay["Wa2"][0]] = False
        assert np.max(np.abs(th_gpu[sel] - o.theta[sel])) < 2e-3 * lr + 1e-7, which
    assert np.max(np.abs(th_gpu - o.theta)) <= 2.1 * 1e-2     # nothing moves more than both lrs allow
    pop.close()


@pytest.mark.parametrize("kernel", KERNELS)
def test_ten_updates_from_replay_with_host_indices(hip_lib, kernel):
    """BaseAgent.learn path: host-drawn (reference RNG) indices -> device gather -> update, ten times.
    Taps of every update stay within 1e-4 of the oracle trajectory (rounding differences compound
    through Adam); the first update is within 1e-5."""
    from oracle.ddpg import Dims, init_params
    from oracle.cpu_baseline import synthetic_pendulum_replay
    dims, B, N = (3, 1, 200, 200, 200), 100, 4096
    pop, _, _, _ = _make(dims, B, cap=N, smin=np.array(SMIN, float), smax=np.array(SMAX, float), amax=AMAX)
    _skip_unless_supported(pop, kernel)
    th = init_params(Dims(*dims), 0)
    pop.set_params(0, th)
    s, a, r, s2, g = synthetic_pendulum_replay(N, 0)
    g = g.copy(); g[::7] = 0.0
    pop.replay_add_batch(0, s, a, r, s2, g)
    o = _oracle(dims, th, (1e-3, 1e-2), SMIN, SMAX, AMAX)
    from rlcontrol_amd.utils.custom_collections import DistinctIndexSampler
    smp = DistinctIndexSampler(0)
    for it in range(10):
        idx = smp.sample_n_k(N, B)
        pop.update(1, host_indices=idx)
        taps = o.update(s[idx], a[idx], s2[idx], r[idx], g[idx], taps=True)
        tol = 1e-5 if it == 0 else 2e-4
        for name in ("q", "y", "a_out", "dqda"):
            assert _rel(pop.last_tap(0, name), taps[name]) < tol, (it, name)
    assert _rel(pop.get_blob(0, "theta_target"), o.theta_t) < 1e-4
    pop.close()


@pytest.mark.parametrize("kernel", KERNELS)
def test_k_updates_in_one_launch_equal_k_launches(hip_lib, kernel):
    """n_updates per launch is only a launch-count choice: bit-identical state either way."""
    from oracle.ddpg import Dims, init_params
    dims, B, N = (3, 1, 200, 200, 200), 100, 2000
    rng = np.random.RandomState(3)
    data = (rng.randn(N, 3), rng.randn(N, 1), rng.randn(N), rng.randn(N, 3), np.full(N, 0.99))
    idx = np.stack([rng.choice(N, B, replace=False) for _ in range(4)]).astype(np.int64)
    blobs = []
    for mode in ("one", "many"):
        pop, _, _, _ = _make(dims, B, cap=N)
        _skip_unless_supported(pop, kernel)
        pop.set_params(0, init_params(Dims(*dims), 5))
        pop.replay_add_batch(0, *data)
        if mode == "one":
            pop.update(4, host_indices=idx)
        else:
            for k in range(4):
                pop.update(1, host_indices=idx[k])
        blobs.append([pop.get_blob(0, w) for w in ("theta", "theta_target", "actor_m", "critic_v")])
        pop.close()
    for x, y in zip(*blobs):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("kernel", KERNELS)
def test_agents_are_independent_and_per_agent_lr(hip_lib, kernel):
    """Population semantics: agent i's result does not depend on its neighbours; each agent uses its own
    learning rates (the INDEX sweep's settings)."""
    from oracle.ddpg import Dims, init_params
    dims, B, N, NA = (3, 1, 200, 200, 200), 100, 1500, 5
    rng = np.random.RandomState(8)
    from rlcontrol_amd.hip_ddpg import DDPGPopulation
    lra = np.array([1e-3, 5e-3, 1e-3, 5e-4, 1e-4], np.float32)
    lrc = np.array([1e-2, 5e-1, 1e-1, 5e-2, 1e-2], np.float32)
    pop = DDPGPopulation(NA, 3, 1, 200, 200, 200, B, N, 0.01, SMIN, SMAX, AMIN, AMAX, lra, lrc,
                         seeds=np.arange(NA) + 1)
    _skip_unless_supported(pop, kernel)
    data = (rng.randn(N, 3), rng.randn(N, 1), rng.randn(N), rng.randn(N, 3), np.full(N, 0.99))
    ths = [init_params(Dims(*dims), 20 + i) for i in range(NA)]
    for i in range(NA):
        pop.set_params(i, ths[i])
        pop.replay_add_batch(i, *data)
    idx = np.stack([rng.choice(N, B, replace=False) for _ in range(NA * 2)]).astype(np.int64).reshape(NA, 2, B)
    pop.update(2, host_indices=idx)
    for i in range(NA):
        o = _oracle(dims, ths[i], (float(lra[i]), float(lrc[i])), SMIN, SMAX, AMAX)
        for k in range(2):
            j = idx[i, k]
            taps = o.update(data[0][j], data[1][j], data[3][j], data[2][j], data[4][j], taps=True)
        for name in ("q", "y", "a_out", "dqda"):
            assert _rel(pop.last_tap(i, name), taps[name]) < 2e-4, (i, name)
    pop.close()


@pytest.mark.parametrize("kernel", KERNELS)
def test_device_sampler_path_runs_and_learns_critic(hip_lib, kernel):
    """Device-Philox sampling (host_indices=None): statistical parity only -- the TD error on a fixed
    probe batch falls when the critic is trained on a stationary synthetic replay."""
    from oracle.ddpg import Dims, init_params
    from oracle.cpu_baseline import synthetic_pendulum_replay
    dims, B, N = (3, 1, 200, 200, 200), 100, 20000
    pop, _, _, _ = _make(dims, B, cap=N, smin=np.array(SMIN, float), smax=np.array(SMAX, float), amax=AMAX,
                         lr=(1e-4, 1e-3))
    _skip_unless_supported(pop, kernel)
    pop.set_params(0, init_params(Dims(*dims), 1))
    s, a, r, s2, g = synthetic_pendulum_replay(N, 0)
    pop.replay_add_batch(0, s, a, r, s2, np.zeros(N))       # gamma_i = 0: y = r, pure regression
    q0 = pop.qval(0, s[:512], a[:512])
    pop.update(300)
    q1 = pop.qval(0, s[:512], a[:512])
    e0, e1 = np.mean((q0 - r[:512]) ** 2), np.mean((q1 - r[:512]) ** 2)
    assert e1 < 0.5 * e0, (e0, e1)
    pop.close()


def test_act_and_qval_match_oracle(hip_lib):
    from oracle.ddpg import Dims, init_params
    for dims in ((3, 1, 200, 200, 200), (8, 2, 64, 48, 40)):
        pop, smin, smax, amax = _make(dims, 8, n_agents=3)
        rng = np.random.RandomState(4)
        for i in range(3):
            th = init_params(Dims(*dims), 30 + i)
            pop.set_params(i, th)
            o = _oracle(dims, th, (1e-3, 1e-2), smin, smax, amax)
            st = rng.uniform(-3, 3, (1, dims[0]))
            assert _rel(pop.act(st, first_agent=i), o.act(st)) < 1e-5
            sts, acts = rng.uniform(-3, 3, (33, dims[0])), rng.uniform(-2, 2, (33, dims[1]))
            assert _rel(pop.qval(i, sts, acts), o.qval(sts, acts)) < 1e-5
        # batched act: one state per agent
        sts = rng.uniform(-3, 3, (3, dims[0]))
        got = pop.act(sts)
        for i in range(3):
            assert np.array_equal(got[i], pop.act(sts[i:i + 1], first_agent=i)[0])
        pop.close()


def test_critic_known_answer_from_reference_checkpoint(hip_lib, golden_dir):
    """The reference's own Bimodal critic checkpoint evaluated by the HIP qval kernel: peaks at a=-1 (1.0)
    and a=+1 (1.5) -- pins layer order, W[in,out] and action-as-last-row on the device."""
    from oracle.ddpg import Dims
    ck = np.load(os.path.join(golden_dir, "bimodal_uneq_var1_qf.npz"))
    dims = (1, 1, 200, 4, 200)
    from rlcontrol_amd.hip_ddpg import DDPGPopulation
    pop = DDPGPopulation(1, 1, 1, 200, 4, 200, 4, 16, 0.01, [-1e9], [1e9], [-1.0], [1.0], 1e-3, 1e-2, seeds=[1],
                         clip_state=False)
    lay, P = Dims(*dims).layout()
    th = np.zeros(P, np.float32)
    for name, src in (("W1", "W1"), ("b1", "b1"), ("Wc2", "W2"), ("bc2", "b2"), ("Wc3", "W3"), ("bc3", "b3")):
        off, shp = lay[name]
        th[off:off + int(np.prod(shp))] = ck[src].reshape(-1)
    pop.set_params(0, th)
    acts = np.linspace(-2, 2, 401)
    q = pop.qval(0, np.zeros((401, 1)), acts[:, None])
    assert abs(acts[acts < 0][np.argmax(q[acts < 0])] + 1.0) < 0.1
    assert abs(acts[acts > 0][np.argmax(q[acts > 0])] - 1.0) < 0.1
    assert abs(q[acts < 0].max() - 1.0) < 0.05 and abs(q[acts > 0].max() - 1.5) < 0.05
    pop.close()


def test_device_ou_noise_statistics(hip_lib):
    """Device OU generator (Philox normals): stationary std sigma/sqrt(1-(1-theta)^2), lag-1 correlation
    1-theta, reset -> mu (utils/exploration_policy.py:18-24).  Statistical parity."""
    from oracle.ddpg import Dims, init_params
    dims = (3, 1, 32, 32, 32)
    from rlcontrol_amd.hip_ddpg import DDPGPopulation
    NA = 64
    pop = DDPGPopulation(NA, 3, 1, 32, 32, 32, 4, 16, 0.01, SMIN, SMAX, [-100.0], [100.0], 1e-3, 1e-2,
                         seeds=np.arange(NA) + 100)
    th = init_params(Dims(*dims), 0)
    for i in range(NA):
        pop.set_params(i, th)
    st = np.tile(np.array([[0.3, -0.2, 1.0]]), (NA, 1))
    greedy = pop.act(st)
    xs = []
    for t in range(400):
        xs.append(pop.act(st, explore=True) - greedy)
    x = np.array(xs)[100:, :, 0]                       # [T, NA]
    std = x.std()
    want = 0.2 / np.sqrt(1 - 0.85 ** 2)
    assert abs(std - want) / want < 0.05
    c = np.mean(x[1:] * x[:-1]) / np.mean(x * x)
    assert abs(c - 0.85) < 0.03
    assert abs(x.mean()) < 0.03
    pop.reset_noise()
    one = pop.act(st, explore=True) - greedy           # first draw after reset: N(0, sigma)
    assert abs(one.std() - 0.2) < 0.08
    pop.close()


@pytest.mark.parametrize("dims,B", [((3, 1, 200, 200, 200), 100), ((8, 2, 64, 48, 40), 17), ((5, 2, 200, 120, 56), 32)])
def test_weight_layout_round_trip_across_kernel_switch(hip_lib, dims, B):
    """The tile-blocked device layout of Wa2 / Wc2 (MFMA kernel) is private to the library: the ABI blob is
    row-major under either kernel, survives switching the kernel (re-pack on the host), and both kernels compute
    the same update from it."""
    from oracle.ddpg import Dims, init_params
    pop, smin, smax, amax = _make(dims, B)
    _skip_unless_supported(pop, "mfma")
    th = init_params(Dims(*dims), 9)
    pop.set_params(0, th)
    rng = np.random.RandomState(2)
    m = rng.randn(th.size).astype(np.float32)
    pop.set_blob(0, "critic_m", m)
    assert np.array_equal(pop.get_blob(0, "theta"), th) and np.array_equal(pop.get_blob(0, "theta_target"), th)
    pop.set_kernel("generic")                      # blocked -> row-major
    assert pop.kernel_in_use() == "generic"
    assert np.array_equal(pop.get_blob(0, "theta"), th) and np.array_equal(pop.get_blob(0, "critic_m"), m)
    s, a, s2, r, g = _batch(rng, B, dims[0], dims[1])
    pop.set_blob(0, "critic_m", np.zeros_like(m))
    pop.update_batch(0, s, a, s2, r, g)
    after_generic = pop.get_blob(0, "theta")
    pop.set_kernel("mfma")                         # row-major -> blocked, optimizer state included
    assert np.array_equal(pop.get_blob(0, "theta"), after_generic)
    pop2, _, _, _ = _make(dims, B)
    pop2.set_kernel("mfma")
    pop2.set_params(0, th)
    pop2.update_batch(0, s, a, s2, r, g)
    assert _rel(pop2.get_blob(0, "theta_target"), pop.get_blob(0, "theta_target")) < 1e-6
    pop.update_batch(0, s, a, s2, r, g)            # second update under the MFMA kernel from the re-packed state
    pop2.update_batch(0, s, a, s2, r, g)
    assert _rel(pop2.get_blob(0, "critic_v"), pop.get_blob(0, "critic_v")) < 1e-4
    assert _rel(pop2.get_blob(0, "theta_target"), pop.get_blob(0, "theta_target")) < 1e-5


@pytest.mark.parametrize("kernel", KERNELS)
def test_device_sampler_updates_equal_oracle_on_the_same_philox_minibatches(hip_lib, kernel):
    """The fully fused path (device sampler -> gather -> update, K updates in one launch) against the oracle fed
    with the minibatches oracle/philox.py draws from the same (seed, call) stream: the indices are bit-identical
    (tests/test_gpu_rollout.py), so the weights after K updates must agree like any other update parity."""
    from oracle import philox
    from oracle.ddpg import Dims, init_params
    from oracle.cpu_baseline import synthetic_pendulum_replay
    dims, B, N, K, seed = (3, 1, 200, 200, 200), 100, 5000, 6, 424242
    pop, smin, smax, amax = _make(dims, B, cap=N, seeds=[seed], smin=np.array(SMIN, float), smax=np.array(SMAX, float),
                                  amax=AMAX)
    _skip_unless_supported(pop, kernel)
    th = init_params(Dims(*dims), 2)
    pop.set_params(0, th)
    s, a, r, s2, g = synthetic_pendulum_replay(N, 0)
    pop.replay_add_batch(0, s, a, r, s2, g)
    o = _oracle(dims, th, (1e-3, 1e-2), smin, smax, amax)
    pop.update(K)
    for call in range(K):
        i = philox.sample_distinct(N, B, seed, call)
        taps = o.update(s[i], a[i], s2[i], r[i], g[i], taps=True)
    assert _rel(pop.last_tap(0, "q"), taps["q"]) < 1e-4            # q of the K-th minibatch after K-1 shared updates
    dev, orc = pop.named(pop.get_blob(0, "theta_target")), pop.named(o.theta_t)
    for name in dev:
        assert _rel(dev[name], orc[name]) < 2e-4, name
    assert np.allclose(pop.get_beta_powers(0), o.pw, rtol=1e-6)
    pop.close()


# ------------------------------------------------------------------ latency mode: one agent over several CUs
SPLIT_CASES = [((3, 1, 200, 200, 200), 100, 4), ((3, 1, 200, 200, 200), 100, 7), ((3, 1, 200, 200, 200), 100, 2),
               ((3, 1, 200, 200, 200), 32, 2), ((8, 2, 200, 160, 144), 64, 4), ((3, 1, 128, 128, 128), 128, 8),
               ((3, 1, 200, 200, 200), 32, 4)]       # the last one leaves two workgroups without rows


@pytest.mark.parametrize("dims,B,C", SPLIT_CASES)
def test_split_update_single_step_matches_oracle(hip_lib, dims, B, C):
    """rlc_ddpg_set_split: the minibatch of one agent over C workgroups, partial gradients reduced over them.
    Same taps and gradients as the oracle at 1e-5 (only the summation order over the batch differs)."""
    from oracle.ddpg import Dims, init_params
    pop, smin, smax, amax = _make(dims, B, kernel="mfma")
    pop.set_split(C)
    pop.enable_grad_taps(True)
    th = init_params(Dims(*dims), 3)
    pop.set_params(0, th)
    o = _oracle(dims, th, (1e-3, 1e-2), smin, smax, amax)
    rng = np.random.RandomState(11)
    s, a, s2, r, g = _batch(rng, B, dims[0], dims[1])
    pop.update_batch(0, s, a, s2, r, g)
    taps = o.update(s, a, s2, r, g, taps=True)
    for name in ("q", "y", "a_out", "dqda"):
        assert _rel(pop.last_tap(0, name), taps[name]) < 1e-5, name
    lay, P = Dims(*dims).layout()
    for which in ("grads_c", "grads_a"):
        got, want = pop.last_tap(0, which), taps[which]
        assert _cos(got, want) > 1 - 1e-9, which
        for name, (off, shp) in lay.items():
            n = int(np.prod(shp))
            if np.any(want[off:off + n]):
                assert _rel(got[off:off + n], want[off:off + n]) < 1e-5, (which, name)
            else:
                assert not np.any(got[off:off + n]), (which, name)
    assert _rel(pop.get_blob(0, "theta_target"), o.theta_t) < 1e-5
    assert np.allclose(pop.get_beta_powers(0), o.pw, rtol=1e-6)
    pop.close()


@pytest.mark.parametrize("C", [2, 4, 7])
def test_split_trajectory_follows_the_one_workgroup_kernel(hip_lib, C):
    """Six updates from the replay (host indices, then the device sampler, several updates per launch, two agents with
    their own learning rates): the split population stays on the one-workgroup MFMA kernel's trajectory."""
    from oracle.ddpg import Dims, init_params
    dims, B, N = (3, 1, 200, 200, 200), 100, 3000
    rng = np.random.RandomState(3)
    data = (rng.randn(N, 3), rng.randn(N, 1), rng.randn(N), rng.randn(N, 3), np.full(N, 0.99))
    idx = np.stack([rng.choice(N, B, replace=False) for _ in range(6)]).reshape(2, 3, B).astype(np.int64)
    pops = []
    for split in (1, C):
        pop, _, _, _ = _make(dims, B, n_agents=2, cap=N, kernel="mfma", lr=([1e-3, 3e-4], [1e-2, 3e-3]))
        pop.set_split(split)
        for i in range(2):
            pop.set_params(i, init_params(Dims(*dims), 5 + i))
            pop.replay_add_batch(i, *data)
        pop.update(3, host_indices=idx)
        pop.update(2)                    # device sampler: every workgroup draws the same index set
        pop.update(1)
        pops.append(pop)
    a, b = pops
    for i in range(2):
        for name in ("q", "y", "a_out", "dqda"):
            assert _rel(b.last_tap(i, name), a.last_tap(i, name)) < 2e-4, (i, name)
        for w in ("theta", "theta_target"):
            assert _rel(b.get_blob(i, w), a.get_blob(i, w)) < 2e-4, (i, w)
        assert np.array_equal(b.get_beta_powers(i), a.get_beta_powers(i))
    # the sampler's stream position advanced identically: one more update draws the same minibatch on both
    for p in pops:
        p.update(1)
    assert _rel(b.last_tap(0, "y"), a.last_tap(0, "y")) < 2e-4
    for p in pops:
        p.close()


def test_split_is_refused_where_it_cannot_run(hip_lib):
    from rlcontrol_amd._lib import RlcError
    pop, _, _, _ = _make((3, 1, 200, 200, 200), 100, n_agents=64, kernel="mfma")
    with pytest.raises(RlcError, match="co-resident"):
        pop.set_split(8)                 # 64 agents x 8 workgroups > 256 CUs
    pop.set_split(4)                     # 64 x 4 = 256: accepted
    pop.set_split(1)
    pop.close()
    pop, _, _, _ = _make((3, 1, 128, 128, 128), 128, kernel="mfma")
    pop.set_split(2)                     # 2 x 64 rows
    pop.close()
    pop, _, _, _ = _make((8, 2, 64, 48, 40), 17, kernel="generic")
    with pytest.raises(RlcError, match="MFMA"):
        pop.set_split(2)
    pop.close()


def test_split_barrier_failure_leaves_the_state_untouched_and_poisons_the_handle(hip_lib):
    """A cross-workgroup barrier that does not complete (test hook: the error word is found set): every workgroup leaves
    at the FIRST barrier, i.e. before the critic reduce / Adam -- parameters, optimizer state and targets keep their
    bits, the call fails, and the handle refuses latency-mode updates until set_split re-arms it."""
    from oracle.ddpg import Dims, init_params
    from rlcontrol_amd._lib import RlcError
    dims, B, N = (3, 1, 200, 200, 200), 100, 2000
    rng = np.random.RandomState(5)
    data = (rng.randn(N, 3), rng.randn(N, 1), rng.randn(N), rng.randn(N, 3), np.full(N, 0.99))
    pop, _, _, _ = _make(dims, B, n_agents=1, cap=N, kernel="mfma")
    pop.set_params(0, init_params(Dims(*dims), 3))
    pop.replay_add_batch(0, *data)
    pop.set_split(4)
    pop.update(2)                                        # a healthy latency-mode launch first
    before = {w: pop.get_blob(0, w) for w in ("theta", "theta_target", "critic_m", "critic_v", "actor_m", "actor_v")}
    pw = pop.get_beta_powers(0)
    pop.debug_fail_next_split()
    with pytest.raises(RlcError, match="did not complete"):
        pop.update(1)
    for w, v in before.items():
        assert np.array_equal(pop.get_blob(0, w), v), w   # nothing was stored behind the failed barrier
    assert np.array_equal(pop.get_beta_powers(0), pw)
    with pytest.raises(RlcError, match="an earlier update of this handle failed"):
        pop.update(1)                                    # poisoned
    pop.set_split(4)                                     # the caller re-arms (state verified above)
    pop.update(1)
    assert not np.array_equal(pop.get_blob(0, "theta"), before["theta"])
    pop.close()
