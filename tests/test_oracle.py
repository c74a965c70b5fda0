"""The oracle must be trustworthy before it judges the HIP path (CPU only).
 (1) two independent restatements agree: hand back-prop fp32 C vs torch.autograd float64;
 (2) known-answer layout check against the critic checkpoint the reference ships
     (Bimodal1DEnv_trueQ_ckpt): layer order, W[in,out], action-as-last-row."""
import os

import numpy as np
import pytest

from oracle.ddpg import DDPGOracle, Dims, init_params
from oracle.cpu_baseline import synthetic_pendulum_replay
from torch_ref import TorchDDPG

SMIN, SMAX, AMAX = [-1, -1, -8], [1, 1, 8], [2.0]


def _rel(x, y):
    x, y = np.asarray(x, np.float64).ravel(), np.asarray(y, np.float64).ravel()
    return np.max(np.abs(x - y)) / (np.max(np.abs(y)) + 1e-30)


@pytest.mark.parametrize("dims,B", [((3, 1, 200, 200, 200), 100), ((3, 1, 200, 200, 200), 32),
                                    ((8, 2, 64, 48, 40), 17), ((1, 1, 16, 16, 16), 5)])
def test_oracle_agrees_with_float64_autograd(dims, B):
    d = Dims(*dims)
    th = init_params(d, 3)
    rng = np.random.RandomState(11)
    smin, smax, amax = -np.ones(d.S) * 2, np.ones(d.S) * 2, np.linspace(1.0, 2.0, d.A)
    o = DDPGOracle(d, th, 1e-3, 1e-2, 0.01, smin, smax, amax)
    t = TorchDDPG(d.tuple(), th, 1e-3, 1e-2, 0.01, smin, smax, amax)
    s = rng.uniform(-3, 3, (B, d.S))        # exercises the state clip (Q6)
    a = rng.uniform(-2, 2, (B, d.A))
    s2 = rng.uniform(-3, 3, (B, d.S))
    r = rng.uniform(-16, 0, B)
    g = np.where(rng.rand(B) < 0.2, 0.0, 0.99)
    to = o.update(s, a, s2, r, g, taps=True)
    tt = t.update(s, a, s2, r, g)
    assert _rel(to["q"], tt["q"]) < 1e-5
    assert _rel(to["y"], tt["y"]) < 1e-6
    assert _rel(to["a_out"], tt["a_out"]) < 1e-5
    assert _rel(to["dqda"], tt["dqda"]) < 1e-5
    lay, _ = d.layout()
    for name, gname in (("grads_c", t.critic_vars), ("grads_a", t.actor_vars)):
        for v in gname:
            off, shp = lay[v]
            n = int(np.prod(shp))
            assert _rel(to[name][off:off + n], tt[name][v]) < 2e-5, (name, v)
    # None-gradient variables stay untouched by the other optimizer (hydra_ddpg_network.py:37,72)
    off, shp = lay["Wa2"]
    assert not np.any(to["grads_c"][off:off + int(np.prod(shp))])
    off, shp = lay["Wc2"]
    assert not np.any(to["grads_a"][off:off + int(np.prod(shp))])
    assert _rel(o.theta_t, t.blob(True)) < 1e-5
    assert np.allclose(o.pw, [0.81, 0.998001, 0.81, 0.998001], rtol=1e-6)


def test_oracle_adam_is_tf_style_not_torch_style():
    """Q2: with epsilon OUTSIDE the bias correction the first step is lr*g/(|g| + eps/sqrt(1-beta2));
    torch.optim.Adam would give lr*g/(|g| + eps)."""
    d = Dims(3, 1, 8, 8, 8)
    th = init_params(d, 0)
    o = DDPGOracle(d, th, 1e-3, 1e-2, 0.01, SMIN, SMAX, AMAX)
    s, a, r, s2, g = synthetic_pendulum_replay(16, 1)
    before = o.theta.copy()
    taps = o.update(s, a, s2, r, g, taps=True)
    lay, _ = d.layout()
    off, shp = lay["Wc2"]
    n = int(np.prod(shp))
    gc = taps["grads_c"][off:off + n]
    step = before[off:off + n] - o.theta[off:off + n]
    big = np.abs(gc) > 1e-5
    assert big.sum() > 10
    tf_style = 1e-2 * gc / (np.abs(gc) + 1e-8 / np.sqrt(1 - 0.999))
    torch_style = 1e-2 * gc / (np.abs(gc) + 1e-8)
    assert np.allclose(step[big], tf_style[big], rtol=2e-4, atol=0)
    small = big & (np.abs(gc) < 1e-3)
    assert small.sum() > 0
    assert np.all(np.abs(step[small] - tf_style[small]) < np.abs(step[small] - torch_style[small]))


def test_oracle_shared_trunk_stepped_by_both_adams():
    """Q1: W1 receives a critic-Adam step and then an actor-Adam step within one update."""
    d = Dims(3, 1, 8, 8, 8)
    th = init_params(d, 0)
    o = DDPGOracle(d, th, 1e-3, 1e-2, 0.01, SMIN, SMAX, AMAX)
    s, a, r, s2, g = synthetic_pendulum_replay(16, 2)
    o.update(s, a, s2, r, g)
    lay, _ = d.layout()
    off, shp = lay["W1"]
    n = int(np.prod(shp))
    assert np.any(o.m_a[off:off + n]) and np.any(o.m_c[off:off + n])
    off, shp = lay["Wc2"]
    assert not np.any(o.m_a[off:off + int(np.prod(shp))])


def test_critic_layout_against_reference_checkpoint(golden_dir):
    """main/qf of Bimodal1DEnv_uneq_var1 approximates the env reward at s=0: peaks near a=-1 (height 1)
    and a=+1 (height 1.5) (environments/environments.py:311-325).  A wrong layer order, a transposed W
    or action-as-first-row would not reproduce them."""
    ck = np.load(os.path.join(golden_dir, "bimodal_uneq_var1_qf.npz"))
    d = Dims(1, 1, 200, 4, 200)          # the actor head is irrelevant for Q; keep it tiny
    lay, P = d.layout()
    th = np.zeros(P, np.float32)
    for name, src in (("W1", "W1"), ("b1", "b1"), ("Wc2", "W2"), ("bc2", "b2"), ("Wc3", "W3"), ("bc3", "b3")):
        off, shp = lay[name]
        th[off:off + int(np.prod(shp))] = ck[src].reshape(-1)
    o = DDPGOracle(d, th, 1e-3, 1e-2, 0.01, [-1e9], [1e9], [1.0], clip_state=False)
    acts = np.linspace(-2, 2, 401)
    q = o.qval(np.zeros((401, 1)), acts[:, None])
    left = acts[acts < 0][np.argmax(q[acts < 0])]
    right = acts[acts > 0][np.argmax(q[acts > 0])]
    assert abs(left + 1.0) < 0.1 and abs(right - 1.0) < 0.1
    assert abs(q[acts < 0].max() - 1.0) < 0.05 and abs(q[acts > 0].max() - 1.5) < 0.05
    # Adam accumulators in the checkpoint (Q2): pinned bit for bit in tests/test_ckpt_pins.py (fp32 running product)
    assert float(ck["beta1_power"]) == 0.0 and np.float32(ck["beta2_power"]) == np.float32(4.5134042e-05)


# ---- the CPU restatements of the on-device experiment loop run on their own (the GPU tests hold the kernels to them) ----
SMIN, SMAX, AMAX = np.array([-1.0, -1.0, -8.0]), np.array([1.0, 1.0, 8.0]), np.array([2.0])


def _loop_invariants(orc, total, interval, episodes, limit):
    assert orc.total == total and len(orc.replay) == total - total // limit         # the truncating step is not stored
    assert orc.timesteps_at_eval == list(range(0, total, interval))
    assert all(len(e) == episodes for e in orc.eval_ret) and np.all(np.isfinite(np.array(orc.eval_ret)))
    assert all(n == limit for n in orc.train_len) and np.all(np.isfinite(np.array(orc.train_ret)))
    acts = np.array([t[1] for t in orc.replay])
    assert np.all(np.abs(acts) <= 2.0) and acts.std() > 0


@pytest.mark.parametrize("norm,sep", [(True, False), (False, True), (True, True)])
def test_variant_rollout_oracle_runs_the_experiment_loop(norm, sep):
    from oracle.ddpg_variants import VDims, init_params
    from oracle.rollout import VariantRolloutOracle
    d = VDims(3, 1, 16, 16, 16, norm=norm, separate=sep)
    orc = VariantRolloutOracle(d, init_params(d, 1), 1e-3, 1e-2, 0.01, SMIN, SMAX, -AMAX, AMAX, 5, 8, 512, 0.99, 0, 20, 70, 30,
                               2).run()
    _loop_invariants(orc, 70, 30, 2, 20)
    assert orc.n_updates >= len(orc.replay) - 8 and not np.array_equal(orc.net.theta, init_params(d, 1))


def test_naf_layer_norm_rollout_oracle_runs_the_experiment_loop():
    from oracle.naf import NafDims
    from oracle.naf_variants import init_params
    from oracle.rollout import NafRolloutOracle
    dims = (3, 1, 16, 16)
    th = init_params(dims, 1, True)
    orc = NafRolloutOracle(NafDims(*dims), th, 1e-3, 0.01, SMIN, SMAX, AMAX, 0.3, 5, 8, 512, 0.99, 0, 20, 70, 30, 2,
                           norm_type="layer").run()
    _loop_invariants(orc, 70, 30, 2, 20)
    assert orc.n_updates >= len(orc.replay) - 8 and not np.array_equal(orc.net.theta.numpy(), th)
