"""Pins taken from the ONLY network-side artefacts the reference holds: the five Bimodal1DEnv_trueQ_ckpt checkpoints
(SAC `main/qf` critics trained for 10 000 Adam steps by TF-1.15; agents/SoftActorCritic.py:37-49 loads them).
tests/golden/make_golden.py parses them weights-only into bimodal_qf_ckpts.npz.

What they pin:
  * the optimizer's beta2 power is the fp32 RUNNING PRODUCT p <- p * 0.999f (TF keeps beta powers as fp32 variables
    multiplied after every apply_gradients), bit for bit -- not 0.999**t in any other arithmetic; the oracle and the
    device kernels accumulate it the same way and land on the same bits after 10 000 updates;
  * value | Adam m | Adam v slot order, W[in, out] layout and action-as-last-row of the concat layer, through the
    exact-zero pattern of the Adam slots (the environment's state is identically 0);
  * layer order / activation placement of the Q network: the checkpoints reproduce the two reward maxima of their
    environments on the C oracle, on the HIP DDPG critic and on the HIP SAC Q path.
"""
import os

import numpy as np
import pytest

NAMES = ["eq_var1", "eq_var2", "eq_var3", "uneq_var1", "uneq_var2"]
STEPS = 10000


@pytest.fixture(scope="module")
def ck(golden_dir):
    return np.load(os.path.join(golden_dir, "bimodal_qf_ckpts.npz"))


FLT_MIN = np.float32(1.17549435e-38)


def _running_product(beta, terms, ftz=True):
    """fp32 accumulator p <- p * beta; ftz: results below the smallest normal flush to zero (TF-1.15 CPU mode)"""
    p = np.float32(beta)
    for _ in range(terms - 1):
        p = np.float32(p * np.float32(beta))
        if ftz and abs(p) < FLT_MIN:
            p = np.float32(0.0)
    return p


def _reward(rew, a):
    m1, m2, h1, h2, s1, s2 = rew
    return h1 * np.exp(-0.5 * ((a - m1) / s1) ** 2) + h2 * np.exp(-0.5 * ((a - m2) / s2) ** 2)


def _q_numpy(ck, name, acts):
    """Q(0, a) of one checkpoint in float64: relu(b1) (the state is 0) -> relu([h, a] W2 + b2) -> W3 + b3"""
    g = lambda t: ck["%s/%s" % (name, t)].astype(np.float64)
    h1 = np.maximum(g("b1"), 0.0)
    h2 = np.maximum(h1 @ g("W2")[:200] + np.outer(acts, g("W2")[200]) + g("b2"), 0.0)
    return h2 @ g("W3")[:, 0] + g("b3")[0]


def _check_maxima(q, acts, rew):
    m1, m2, h1, h2 = rew[:4]
    left, right = acts < 0, acts > 0
    assert abs(acts[left][np.argmax(q[left])] - m1) < 0.1 and abs(acts[right][np.argmax(q[right])] - m2) < 0.1
    assert abs(q[left].max() - h1) < 0.06 and abs(q[right].max() - h2) < 0.06
    assert np.max(np.abs(q - _reward(rew, acts))) < 0.15          # the critic fits the whole reward curve on [-2, 2]


# ---------------------------------------------------------------------------------------------- CPU
@pytest.mark.parametrize("name", NAMES)
def test_beta_powers_are_fp32_running_products(ck, name):
    b1p, b2p = ck[name + "/beta_powers"]
    want = _running_product(0.999, STEPS + 1)           # initial value beta2, then one multiplication per step
    assert want.tobytes() == np.float32(b2p).tobytes()
    # the alternatives a restatement might have used do NOT give these bits
    assert np.float32(np.float32(0.999) ** np.float32(STEPS + 1)).tobytes() != np.float32(b2p).tobytes()
    assert np.float32(0.999 ** (STEPS + 1)).tobytes() != np.float32(b2p).tobytes()
    assert _running_product(0.999, STEPS).tobytes() != np.float32(b2p).tobytes()
    # beta1: 0.9^10001 underflows.  With gradual underflow the fp32 product would stick at 4 ulp (5.6e-45) for ever;
    # the checkpoint holds exactly 0.0 -- TF-1.15 ran with denormals flushed to zero (oracle/ftz.h)
    assert float(b1p) == 0.0 and float(_running_product(0.9, STEPS + 1)) == 0.0
    assert float(_running_product(0.9, STEPS + 1, ftz=False)) > 0.0


def test_oracle_accumulates_beta_powers_like_the_checkpoint(ck):
    """oracle/sac_oracle.c after 10 000 updates holds the checkpoint's beta2 power bit for bit (both optimizers)."""
    from oracle.sac import SACOracle, SacDims, init_params
    d = SacDims(1, 1, 4, 4, 4, 4)
    o = SACOracle(d, init_params(d, 0), 1e-4, 1e-4, 0.1, 0.01, -1.0, 1.0, 1.0)
    rng = np.random.RandomState(0)
    s, a, s2, r, g, eps = rng.randn(2, 1), rng.randn(2, 1), rng.randn(2, 1), rng.randn(2), np.full(2, 0.99), rng.randn(2, 1)
    for _ in range(STEPS):
        o.update(s, a, s2, r, g, eps)
    want = np.float32(ck["uneq_var1/beta_powers"][1]).tobytes()
    assert np.float32(o.pw[1]).tobytes() == want and np.float32(o.pw[3]).tobytes() == want
    assert float(o.pw[0]) == 0.0 and float(o.pw[2]) == 0.0


@pytest.mark.parametrize("name", NAMES)
def test_adam_slot_order_and_weight_layout_from_the_zero_pattern(ck, name):
    g = lambda t: ck["%s/%s" % (name, t)]
    for t in ("b1", "W1", "b2", "W2", "b3", "W3"):
        assert np.all(g(t + "_v") >= 0.0), t                       # second-moment slot really is v
        assert np.all((g(t + "_m") == 0.0) | (g(t + "_v") > 0.0)), t          # m != 0 implies v > 0
    # the state is identically 0: dL/dW1 = 0 * dh1 exactly, for 10 000 steps
    assert np.all(g("W1_m") == 0.0) and np.all(g("W1_v") == 0.0)
    # a first-layer unit that never received gradient (v == 0) was never active: ITS ROW of W2[in, out] is untouched,
    # while the action row (the LAST row of the concat layer) is trained wherever the second layer is alive
    dead1 = g("b1_v") == 0.0
    alive2 = g("b2_v") > 0.0
    assert 20 < dead1.sum() < 180 and alive2.sum() > 100
    W2v = g("W2_v")
    assert np.all(W2v[:200][dead1] == 0.0) and np.all(g("W2_m")[:200][dead1] == 0.0)
    assert np.all(W2v[200][alive2] > 0.0)
    assert np.all(W2v[:, ~alive2] == 0.0) and np.all(g("W3_v")[~alive2, 0] == 0.0)
    # rows of live first-layer units are trained in (nearly) every live column; read as [out, in] the same bytes give
    # 38-48 % here and only 52-60 % zeros in the "dead" rows above
    assert np.mean(W2v[:200][~dead1][:, alive2] > 0.0) > 0.95
    T = W2v.ravel().reshape(200, 201).T
    assert np.mean(T[:200][~dead1][:, alive2] > 0.0) < 0.6 and np.mean(T[:200][dead1] == 0.0) < 0.7
    assert float(g("b3_v")[0]) > 0.0


@pytest.mark.parametrize("name", NAMES)
def test_adam_first_moment_stalls_where_its_decrement_flushes(ck, name):
    """ApplyAdam writes m += (g - m) * (1 - beta1).  Once g is 0 for good, m decays by 0.9 per step until
    m * 0.1 is denormal and flushes: every idle slot of the checkpoints rests in [0.9, 1) x 10 FLT_MIN, none below,
    none denormal.  (m = beta1 * m + (1 - beta1) * g, or arithmetic with gradual underflow, would leave denormals or
    zeros instead.)"""
    lo, hi = np.float32(9.0) * FLT_MIN * np.float32(0.99999), np.float32(10.0) * FLT_MIN
    idle = 0
    for t in ("b1", "b2", "W2", "W3"):
        m = np.abs(ck["%s/%s_m" % (name, t)].ravel())
        assert not np.any((m > 0) & (m < lo)), t
        idle += int(np.sum((m >= lo) & (m < hi)))
    assert idle > 1000


def test_oracle_first_moment_stalls_like_the_checkpoints():
    """the same signature from oracle/sac_oracle.c: a few updates with a non-zero state, then the state goes to 0 for
    good (dL/dqW1 = 0 exactly): the first-layer m slots must come to rest in the checkpoints' band, not underflow"""
    from oracle.sac import SACOracle, SacDims, init_params
    d = SacDims(1, 1, 4, 4, 8, 8)
    o = SACOracle(d, init_params(d, 3), 1e-3, 1e-3, 0.1, 0.01, -5.0, 5.0, 1.0)
    rng = np.random.RandomState(1)
    B = 4
    mk = lambda s: (s, rng.randn(B, 1), s, rng.randn(B), np.full(B, 0.99), rng.randn(B, 1))
    for _ in range(3):
        o.update(*mk(rng.randn(B, 1)))
    lay, _ = d.layout()
    off = lay["qW1"][0]
    assert np.any(o.m[off:off + 8] != 0.0)
    for _ in range(1200):
        o.update(*mk(np.zeros((B, 1))))
    m = np.abs(o.m[off:off + 8])
    lo, hi = np.float32(9.0) * FLT_MIN * np.float32(0.99999), np.float32(10.0) * FLT_MIN
    assert np.all((m == 0.0) | ((m >= lo) & (m < hi))), m
    assert np.any(m > 0.0)


@pytest.mark.parametrize("name", NAMES)
def test_checkpoints_reproduce_the_reward_maxima(ck, name):
    acts = np.linspace(-2, 2, 401)
    _check_maxima(_q_numpy(ck, name, acts), acts, ck[name + "/reward"])


@pytest.mark.parametrize("name", NAMES)
def test_oracle_sac_q_path_on_the_checkpoints(ck, name):
    """oracle/sac_oracle.c with the checkpoint as main/qf: its q tap equals the float64 forward and finds the maxima."""
    from oracle.sac import SACOracle, SacDims
    d = SacDims(1, 1, 8, 8, 200, 200)
    lay, P = d.layout()
    th = np.zeros(P, np.float32)
    th[lay["pWs"][0]:lay["pWs"][0] + 8] = 0.01
    for dst, src in (("qW1", "W1"), ("qb1", "b1"), ("qW2", "W2"), ("qb2", "b2"), ("qW3", "W3"), ("qb3", "b3")):
        off, shp = lay[dst]
        th[off:off + int(np.prod(shp))] = ck["%s/%s" % (name, src)].reshape(-1)
    acts = np.linspace(-2, 2, 101)
    o = SACOracle(d, th, 0.0, 0.0, 0.1, 0.01, -1.0, 1.0, 1.0)
    t = o.update(np.zeros((101, 1)), acts[:, None], np.zeros((101, 1)), np.zeros(101), np.zeros(101), np.zeros((101, 1)),
                 taps=True)
    want = _q_numpy(ck, name, acts)
    assert np.max(np.abs(t["q"] - want)) < 1e-5 * np.max(np.abs(want))
    _check_maxima(t["q"].astype(np.float64), acts, ck[name + "/reward"])


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["generic", "mfma"])
@pytest.mark.parametrize("name", NAMES)
def test_hip_sac_q_path_on_the_checkpoints(hip_lib, ck, name, kernel):
    """HIP SAC kernels (both) with the checkpoint loaded as main/qf: the q tap of an update on (s = 0, a = grid)."""
    from oracle.sac import SacDims
    from rlcontrol_amd.hip_sac import SACPopulation
    B = 64
    pop = SACPopulation(1, 1, 1, 16, 16, 200, 200, B, 256, 0.01, -1.0, 1.0, 1.0, 0.0, 0.0, 0.1, seeds=[1])
    pop.set_kernel(kernel)
    d = SacDims(1, 1, 16, 16, 200, 200)
    lay, P = d.layout()
    th = np.zeros(P, np.float32)
    th[lay["pWs"][0]:lay["pWs"][0] + 16] = 0.01
    for dst, src in (("qW1", "W1"), ("qb1", "b1"), ("qW2", "W2"), ("qb2", "b2"), ("qW3", "W3"), ("qb3", "b3")):
        off, shp = lay[dst]
        th[off:off + int(np.prod(shp))] = ck["%s/%s" % (name, src)].reshape(-1)
    pop.set_params(0, th)
    qs, grid = [], np.linspace(-2, 2, 4 * B)
    for part in range(4):
        acts = grid[part::4]
        pop.update_batch(0, np.zeros((B, 1)), acts[:, None], np.zeros((B, 1)), np.zeros(B), np.zeros(B), eps=np.zeros((B, 1)))
        qs.append((acts, pop.last_tap(0, "q").astype(np.float64)))
    acts = np.concatenate([a for a, _ in qs])
    q = np.concatenate([v for _, v in qs])
    order = np.argsort(acts)
    acts, q = acts[order], q[order]
    want = _q_numpy(ck, name, acts)
    assert np.max(np.abs(q - want)) < 1e-5 * np.max(np.abs(want))
    _check_maxima(q, acts, ck[name + "/reward"])
    pop.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_ddpg_critic_on_the_checkpoints(hip_lib, ck, name):
    """the DDPG qval kernel (hydra critic = the same 1 -> 200 -> [h, a] -> 200 -> 1 stack) on every checkpoint"""
    from oracle.ddpg import Dims
    from rlcontrol_amd.hip_ddpg import DDPGPopulation
    pop = DDPGPopulation(1, 1, 1, 200, 4, 200, 4, 16, 0.01, [-1e9], [1e9], [-1.0], [1.0], 1e-3, 1e-2, seeds=[1],
                         clip_state=False)
    lay, P = Dims(1, 1, 200, 4, 200).layout()
    th = np.zeros(P, np.float32)
    for dst, src in (("W1", "W1"), ("b1", "b1"), ("Wc2", "W2"), ("bc2", "b2"), ("Wc3", "W3"), ("bc3", "b3")):
        off, shp = lay[dst]
        th[off:off + int(np.prod(shp))] = ck["%s/%s" % (name, src)].reshape(-1)
    pop.set_params(0, th)
    acts = np.linspace(-2, 2, 401)
    q = pop.qval(0, np.zeros((401, 1)), acts[:, None]).astype(np.float64)
    want = _q_numpy(ck, name, acts)
    assert np.max(np.abs(q - want)) < 1e-5 * np.max(np.abs(want))
    _check_maxima(q, acts, ck[name + "/reward"])
    pop.close()


@pytest.mark.gpu
def test_device_beta_powers_after_10000_updates_equal_the_checkpoint_bits(hip_lib, ck):
    """the kernels keep beta powers as fp32 accumulators multiplied once per update (quirk Q2): after 10 000 updates in
    a handful of launches they hold the bit pattern TF left in the reference's checkpoints -- SAC (both kernels),
    DDPG (both kernels) and NAF"""
    want = np.float32(ck["uneq_var1/beta_powers"][1]).tobytes()
    rng = np.random.RandomState(0)
    from rlcontrol_amd.hip_sac import SACPopulation, init_params as sac_init
    from rlcontrol_amd.hip_ddpg import DDPGPopulation, init_params as ddpg_init
    from rlcontrol_amd.hip_naf import NAFPopulation, init_params as naf_init
    N, B = 64, 4
    data = (rng.uniform(-1, 1, (N, 3)), rng.uniform(-1, 1, (N, 1)), rng.uniform(-1, 0, N), rng.uniform(-1, 1, (N, 3)),
            np.full(N, 0.99))
    for kernel in ("generic", "mfma"):
        sac = SACPopulation(1, 3, 1, 16, 16, 16, 16, B, N, 0.01, -1.0, 1.0, 1.0, 1e-5, 1e-5, 0.1, seeds=[3])
        sac.set_kernel(kernel)
        sac.set_params(0, sac_init(3, 1, 16, 16, 16, 16, 0))
        sac.replay_add_batch(0, *data)
        for _ in range(10):
            sac.update(STEPS // 10)
        pw = sac.get_beta_powers(0)
        assert np.float32(pw[1]).tobytes() == want and np.float32(pw[3]).tobytes() == want, kernel
        assert pw[0] == 0.0 and pw[2] == 0.0
        sac.close()
        dd = DDPGPopulation(1, 3, 1, 16, 16, 16, B, N, 0.01, [-1, -1, -1], [1, 1, 1], [-1.0], [1.0], 1e-5, 1e-5, seeds=[3])
        dd.set_kernel(kernel)
        dd.set_params(0, ddpg_init(3, 1, 16, 16, 16, 0))
        dd.replay_add_batch(0, *data)
        for _ in range(10):
            dd.update(STEPS // 10)
        pw = dd.get_beta_powers(0)
        assert np.float32(pw[1]).tobytes() == want and np.float32(pw[3]).tobytes() == want, kernel
        dd.close()
        nf = NAFPopulation(1, 3, 1, 16, 16, B, N, 0.01, -np.ones(3), np.ones(3), np.ones(1), 1e-5, seeds=[3])
        nf.set_kernel(kernel)
        nf.set_params(0, naf_init(3, 1, 16, 16, 0))
        nf.replay_add_batch(0, *data)
        for _ in range(10):
            nf.update(STEPS // 10)
        assert np.float32(nf.get_beta_powers(0)[1]).tobytes() == want, kernel
        nf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["generic", "mfma"])
def test_device_first_moment_stalls_like_the_checkpoints(hip_lib, kernel):
    """the stalled-m signature (see the CPU tests above) from the HIP SAC kernels: non-zero states for a few updates,
    then zero states for good: the first-layer m slots of Q rest in [0.9, 1) x 10 FLT_MIN, none denormal"""
    from oracle.sac import SacDims
    from rlcontrol_amd.hip_sac import SACPopulation, init_params
    B = 4
    pop = SACPopulation(1, 1, 1, 16, 16, 16, 16, B, 64, 0.01, -5.0, 5.0, 1.0, 1e-3, 1e-3, 0.1, seeds=[2])
    pop.set_kernel(kernel)
    pop.set_params(0, init_params(1, 1, 16, 16, 16, 16, 3))
    rng = np.random.RandomState(1)
    mk = lambda s: (s, rng.randn(B, 1), s, rng.randn(B), np.full(B, 0.99))
    for _ in range(3):
        pop.update_batch(0, *mk(rng.randn(B, 1)), eps=rng.randn(B, 1))
    lay, _ = SacDims(1, 1, 16, 16, 16, 16).layout()
    off = lay["qW1"][0]
    assert np.any(pop.get_blob(0, "adam_m")[off:off + 16] != 0.0)
    for _ in range(1200):
        pop.update_batch(0, *mk(np.zeros((B, 1))), eps=rng.randn(B, 1))
    m = np.abs(pop.get_blob(0, "adam_m")[off:off + 16])
    lo, hi = np.float32(9.0) * FLT_MIN * np.float32(0.99999), np.float32(10.0) * FLT_MIN
    assert np.all((m == 0.0) | ((m >= lo) & (m < hi))), m
    assert np.any(m > 0.0)
    pop.close()
