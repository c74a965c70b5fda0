"""DDPG with norm_type 'layer' and with separate actor / critic networks.
CPU: the C restatement (oracle/ddpg_variants_oracle.c) reduces to the pinned hydra oracle bit for bit and agrees with a
float64 autograd twin; GPU: the HIP kernel against the oracle at 1e-5 (q, y, a_out, dQ/da, every gradient tensor)."""
import numpy as np
import pytest

from oracle.ddpg_variants import DDPGVariantOracle, VDims, init_params

SMIN, SMAX, AMAX = [-1, -1, -8], [1, 1, 8], [2.0]
VARIANTS = [(True, False), (False, True), (True, True)]
SHAPES = [((3, 1, 200, 200, 200), 100), ((3, 1, 32, 24, 40), 17), ((5, 2, 48, 64, 32), 32)]


def _rel(x, y):
    x, y = np.asarray(x, np.float64).ravel(), np.asarray(y, np.float64).ravel()
    return float(np.max(np.abs(x - y)) / (np.max(np.abs(y)) + 1e-30))


def _bounds(S, A):
    return -np.ones(S) * 2, np.ones(S) * 2, np.linspace(1.0, 2.0, A)


def _batch(rng, B, S, A):
    return (rng.uniform(-3, 3, (B, S)), rng.uniform(-2, 2, (B, A)), rng.uniform(-3, 3, (B, S)),
            rng.uniform(-16, 0, B), np.where(rng.rand(B) < 0.2, 0.0, 0.99))


def test_variant_oracle_reduces_to_the_hydra_oracle_bit_for_bit():
    from oracle.ddpg import DDPGOracle, Dims, init_params as hydra_init
    from oracle.cpu_baseline import synthetic_pendulum_replay
    th = hydra_init(Dims(3, 1, 40, 32, 48), 5)
    a = DDPGOracle(Dims(3, 1, 40, 32, 48), th, 1e-3, 1e-2, 0.01, SMIN, SMAX, AMAX)
    b = DDPGVariantOracle(VDims(3, 1, 40, 32, 48), th, 1e-3, 1e-2, 0.01, SMIN, SMAX, AMAX)
    s, act, r, s2, g = synthetic_pendulum_replay(64, 3)
    for k in range(4):
        i = slice(16 * k, 16 * k + 16)
        ta = a.update(s[i], act[i], s2[i], r[i], g[i], taps=True)
        tb = b.update(s[i], act[i], s2[i], r[i], g[i], taps=True)
        for name in ta:
            assert np.array_equal(ta[name], tb[name]), (k, name)
    for name in ("theta", "theta_t", "m_a", "v_a", "m_c", "v_c", "pw"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    st = np.random.RandomState(0).uniform(-1, 1, (5, 3))
    assert np.array_equal(a.act(st), b.act(st))


def test_variant_layouts():
    assert VDims(3, 1, 200, 200, 200).P == 81802
    assert VDims(3, 1, 200, 200, 200, norm=True).P == 81802 + 2 * 600              # beta + gamma of three layers
    assert VDims(3, 1, 200, 200, 200, separate=True).P == 81802 + 800              # the critic's own 3 -> 200 layer
    lay, _ = VDims(3, 1, 8, 8, 8, norm=True, separate=True).layout()
    assert list(lay) == ["W1", "b1", "l1b", "l1g", "Wa2", "ba2", "l2b", "l2g", "Wa3", "ba3", "Wc1", "bc1", "lcb", "lcg",
                         "Wc2", "bc2", "l3b", "l3g", "Wc3", "bc3"]


@pytest.mark.parametrize("norm,sep", VARIANTS)
@pytest.mark.parametrize("dims,B", SHAPES)
def test_variant_oracle_agrees_with_float64_autograd(dims, B, norm, sep):
    from torch_ref_variants import TorchDDPGVariant
    d = VDims(*dims, norm=norm, separate=sep)
    th = init_params(d, 2)
    rng = np.random.RandomState(1)
    lay, _ = d.layout()
    for n, (off, shp) in lay.items():                       # move gamma / beta off their trivial initial values
        if n[0] == "l":
            k = int(np.prod(shp))
            th[off:off + k] += rng.uniform(-0.3, 0.3, k).astype(np.float32)
    smin, smax, amax = _bounds(dims[0], dims[1])
    o = DDPGVariantOracle(d, th, 1e-3, 1e-2, 0.01, smin, smax, amax)
    t = TorchDDPGVariant(d, th, 1e-3, 1e-2, 0.01, smin, smax, amax)
    for it in range(2):
        s, a, s2, r, g = _batch(rng, B, dims[0], dims[1])
        to = o.update(s, a, s2, r, g, taps=True)
        tt = t.update(s, a, s2, r, g)
        for k in ("q", "y", "a_out", "dqda"):
            assert _rel(to[k], tt[k]) < 2e-5, (it, k)
        if it == 0:
            for tag in ("grads_c", "grads_a"):
                for n, gt in tt[tag].items():
                    off, shp = lay[n]
                    assert _rel(to[tag][off:off + int(np.prod(shp))], gt) < 5e-5, (tag, n)
    assert _rel(o.theta, t.blob()) < 2e-5 and _rel(o.theta_t, t.blob(True)) < 2e-5


def test_layer_norm_statistics_follow_tf_contrib():
    """rows are normalised over the FEATURES with the biased variance and eps = 1e-12; gamma / beta act after"""
    d = VDims(3, 1, 16, 16, 16, norm=True)
    th = init_params(d, 0)
    lay, _ = d.layout()
    o = DDPGVariantOracle(d, th, 0.0, 0.0, 0.0, SMIN, SMAX, AMAX)
    s = np.random.RandomState(2).uniform(-1, 1, (4, 3)).astype(np.float32)
    W1 = th[lay["W1"][0]:lay["W1"][0] + 48].reshape(3, 16).astype(np.float64)
    b1 = th[lay["b1"][0]:lay["b1"][0] + 16].astype(np.float64)
    z = s.astype(np.float64) @ W1 + b1
    n1 = (z - z.mean(1, keepdims=True)) / np.sqrt(z.var(1, keepdims=True) + 1e-12)
    h1 = np.maximum(n1, 0.0)
    Wa2 = th[lay["Wa2"][0]:lay["Wa2"][0] + 256].reshape(16, 16).astype(np.float64)
    z2 = h1 @ Wa2 + th[lay["ba2"][0]:lay["ba2"][0] + 16]
    h2 = np.maximum((z2 - z2.mean(1, keepdims=True)) / np.sqrt(z2.var(1, keepdims=True) + 1e-12), 0.0)
    mu = np.tanh(h2 @ th[lay["Wa3"][0]:lay["Wa3"][0] + 16].reshape(16, 1) + th[lay["ba3"][0]]) * 2.0
    assert _rel(o.act(s), mu) < 1e-5
