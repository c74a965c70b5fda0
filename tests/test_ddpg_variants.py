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


# ------------------------------------------------------------------------------------------ GPU
def _pop(dims, B, norm, sep, n_agents=1, cap=512):
    from rlcontrol_amd.hip_ddpg import DDPGPopulation
    S, A, H1, HA, HC = dims
    smin, smax, amax = _bounds(S, A)
    return DDPGPopulation(n_agents, S, A, H1, HA, HC, B, cap, 0.01, smin, smax, -amax, amax, 1e-3, 1e-2,
                          seeds=list(range(7, 7 + n_agents)), norm_type="layer" if norm else "input_norm",
                          separate_networks=sep)


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["generic", "mfma"])
@pytest.mark.parametrize("norm,sep", VARIANTS)
@pytest.mark.parametrize("dims,B", SHAPES)
def test_hip_variant_update_matches_oracle(hip_lib, dims, B, norm, sep, kernel):
    if kernel == "mfma" and norm:
        pytest.skip("layer norm runs on the any-shape kernel only (DESIGN.md 9)")
    d = VDims(*dims, norm=norm, separate=sep)
    th = init_params(d, 2)
    rng = np.random.RandomState(1)
    lay, P = d.layout()
    for n, (off, shp) in lay.items():
        if n[0] == "l":
            k = int(np.prod(shp))
            th[off:off + k] += rng.uniform(-0.3, 0.3, k).astype(np.float32)
    smin, smax, amax = _bounds(dims[0], dims[1])
    pop = _pop(dims, B, norm, sep)
    # layer norm runs on the any-shape kernel; `network: separate` alone takes the MFMA kernel by default (round 3)
    assert pop.P == P and pop.kernel_in_use() == ("generic" if norm else "mfma")
    pop.set_kernel(kernel)
    assert pop.kernel_in_use() == kernel
    from rlcontrol_amd.hip_ddpg import param_layout
    assert list(param_layout(*dims, "layer" if norm else "input_norm", sep)[0]) == list(lay)
    pop.enable_grad_taps(True)
    pop.set_params(0, th)
    o = DDPGVariantOracle(d, th, 1e-3, 1e-2, 0.01, smin, smax, amax)
    for it in range(3):
        s, a, s2, r, g = _batch(rng, B, dims[0], dims[1])
        pop.update_batch(0, s, a, s2, r, g)
        t = o.update(s, a, s2, r, g, taps=True)
        tol = 1e-5 if it == 0 else 2e-4
        for k in ("q", "y", "a_out", "dqda"):
            assert _rel(pop.last_tap(0, k), t[k]) < tol, (it, k)
        if it == 0:
            for tag in ("grads_c", "grads_a"):
                got = pop.last_tap(0, tag)
                for n, (off, shp) in lay.items():
                    k = int(np.prod(shp))
                    if np.max(np.abs(t[tag][off:off + k])) > 0:
                        assert _rel(got[off:off + k], t[tag][off:off + k]) < 3e-5, (tag, n)
            assert _rel(pop.get_blob(0, "theta"), o.theta) < 1e-5
            assert _rel(pop.get_blob(0, "theta_target"), o.theta_t) < 1e-5
    st = rng.uniform(-2, 2, (1, dims[0]))
    assert _rel(pop.act(st), o.act(st)) < 1e-5                        # acting path (B = 1) through the layer norms
    pop.close()


@pytest.mark.gpu
def test_hip_variant_replay_path_and_kernel_guard(hip_lib):
    """fused sample + gather + update on the replay with layer norm; the MFMA kernel refuses (the device loop runs it:
    tests/test_gpu_rollout.py)"""
    from rlcontrol_amd._lib import RlcError
    dims, B, N = (3, 1, 200, 200, 200), 100, 512
    d = VDims(*dims, norm=True)
    smin, smax, amax = _bounds(3, 1)
    pop = _pop(dims, B, True, False, n_agents=2, cap=N)
    rng = np.random.RandomState(3)
    data = (rng.uniform(-2, 2, (N, 3)), rng.uniform(-1, 1, (N, 1)), rng.uniform(-16, 0, N), rng.uniform(-2, 2, (N, 3)),
            np.full(N, 0.99))
    ths = [init_params(d, 30 + i) for i in range(2)]
    oracles = []
    for i in range(2):
        pop.set_params(i, ths[i])
        pop.replay_add_batch(i, *data)
        oracles.append(DDPGVariantOracle(d, ths[i], 1e-3, 1e-2, 0.01, smin, smax, amax))
    idx = np.stack([rng.choice(N, B, replace=False) for _ in range(4)]).reshape(2, 2, B).astype(np.int64)
    pop.update(2, host_indices=idx)
    for i in range(2):
        for k in range(2):
            j = idx[i, k]
            t = oracles[i].update(data[0][j], data[1][j], data[3][j], data[2][j], data[4][j], taps=True)
        for name in ("q", "y", "dqda"):
            assert _rel(pop.last_tap(i, name), t[name]) < 2e-4, (i, name)
    with pytest.raises(RlcError):
        pop.set_kernel("mfma")
    pop.update(3)                                                      # device sampler path
    assert np.all(np.isfinite(pop.get_blob(1, "theta")))
    pop.close()


@pytest.mark.gpu
def test_dropin_agent_accepts_layer_norm_and_rejects_batch_norm(hip_lib):
    from rlcontrol_amd.utils.config import Config
    from rlcontrol_amd.utils.main_utils import create_agent
    from rlcontrol_amd.environments.environments import create_environment
    env = create_environment({"environment": "Pendulum-v0", "TotalMilSteps": 0.001, "EpisodeSteps": -1,
                              "EvalIntervalMilSteps": 0.0005, "EvalEpisodes": 2})

    def cfg(norm):
        c = Config()
        c.merge_config({"env_name": env.name, "state_dim": env.state_dim, "state_min": env.state_min,
                        "state_max": env.state_max, "action_dim": env.action_dim, "action_min": env.action_min,
                        "action_max": env.action_max, "norm_type": norm, "exploration_policy": "ou_noise",
                        "shared_l1_dim": 64, "actor_l2_dim": 64, "critic_l2_dim": 64, "actor_lr": 1e-3, "critic_lr": 1e-2,
                        "buffer_size": 2000, "writer": None, "write_log": False, "write_plot": False, "random_seed": 0})
        return c
    agent = create_agent("DDPG", cfg("layer"))
    env.set_random_seed(0)
    obs = env.reset()
    agent.reset()
    a = agent.start(obs, True)
    for t in range(60):
        obs_n, r, done, _ = env.step(a)
        agent.update(obs, obs_n, float(r), a, done, False)
        a = agent.step(obs_n, True)
        obs = obs_n
        assert a.shape == (1,) and abs(a[0]) <= 2.0
    with pytest.raises(ValueError):
        create_agent("DDPG", cfg("batch"))
    # SoftActorCritic and NAF take 'layer' too (tests/test_sac.py, tests/test_naf.py); nobody runs 'batch' as input_norm
    for norm in ("layer", "batch"):
        c = cfg(norm)
        c.merge_config({"l1_dim": 16, "l2_dim": 16, "noise_scale": 0.3, "learning_rate": 1e-3,
                        "exploration_policy": "none"})
        if norm == "batch":
            with pytest.raises(ValueError):
                create_agent("NAF", c)
        else:
            naf = create_agent("NAF", c)
            naf.reset()
            assert naf.start(obs, False).shape == (1,) and naf.start(obs, True).shape == (1,)
    c = cfg("batch")
    c.merge_config({"actor_l1_dim": 16, "actor_l2_dim": 16, "critic_l1_dim": 16, "critic_l2_dim": 16, "pi_lr": 1e-3,
                    "qf_vf_lr": 1e-3, "entropy_scale": 0.1, "sample_for_eval": "False", "use_true_q": "False",
                    "exploration_policy": "none"})
    with pytest.raises(ValueError):
        create_agent("SoftActorCritic", c)
