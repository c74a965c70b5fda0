"""NAF: oracle trustworthiness on CPU + HIP parity on the GPU (1e-5 on Q, y, V and gradients)."""
import numpy as np
import pytest

from oracle.naf import NAFOracle, NafDims, init_params

CASES = [((8, 2, 200, 200), 32), ((3, 1, 64, 48), 17), ((5, 3, 32, 40), 9), ((8, 2, 200, 200), 100)]


def _rel(x, y):
    x, y = np.asarray(x, np.float64).ravel(), np.asarray(y, np.float64).ravel()
    return float(np.max(np.abs(x - y)) / (np.max(np.abs(y)) + 1e-30))


def _bounds(S, A):
    return -np.ones(S) * 2, np.ones(S) * 2, np.linspace(1.0, 2.0, A)


def _batch(rng, B, S, A):
    return (rng.uniform(-3, 3, (B, S)), rng.uniform(-2, 2, (B, A)), rng.uniform(-3, 3, (B, S)),
            rng.uniform(-16, 0, B), np.where(rng.rand(B) < 0.2, 0.0, 0.99))


def test_naf_param_count_matches_survey():
    assert NafDims(8, 2, 200, 200).P == 83406


@pytest.mark.parametrize("dims,B", CASES)
def test_naf_oracle_agrees_with_float64_autograd(dims, B):
    from torch_ref_naf import TorchNAF
    d = NafDims(*dims)
    th = init_params(d, 2)
    smin, smax, amax = _bounds(dims[0], dims[1])
    o = NAFOracle(d, th, 1e-3, 0.01, smin, smax, amax)
    t = TorchNAF(dims, th, 1e-3, 0.01, smin, smax, amax)
    rng = np.random.RandomState(1)
    s, a, s2, r, g = _batch(rng, B, dims[0], dims[1])
    to = o.update(s, a, s2, r, g, taps=True)
    tt = t.update(s, a, s2, r, g)
    for k in ("q", "y", "V"):
        assert _rel(to[k], tt[k]) < 1e-5, k
    lay, _ = d.layout()
    for n, (off, shp) in lay.items():
        assert _rel(to["grads"][off:off + int(np.prod(shp))], tt["grads"][n]) < 2e-5, n
    assert _rel(o.theta_t, t.blob(True)) < 1e-5


def test_naf_loss_is_a_sum_not_a_mean():
    """naf_network.py:53: dQ seed is 2(q - y), independent of the batch size."""
    d = NafDims(3, 1, 16, 16)
    th = init_params(d, 0)
    smin, smax, amax = _bounds(3, 1)
    rng = np.random.RandomState(0)
    s, a, s2, r, g = _batch(rng, 8, 3, 1)
    o = NAFOracle(d, th, 1e-3, 0.01, smin, smax, amax)
    t = o.update(s, a, s2, r, g, taps=True)
    lay, _ = d.layout()
    assert abs(t["grads"][lay["bv3"][0]] - np.sum(2.0 * (t["q"] - t["y"]))) < 1e-4


# ------------------------------------------------------------------------------------------ GPU
KERNELS = ["generic", "mfma"]


def _pop(dims, B, n_agents=1, cap=2048, lr=1e-3, kernel="auto"):
    from rlcontrol_amd.hip_naf import NAFPopulation
    S, A, L1, L2 = dims
    if kernel == "mfma" and A > 2:
        pytest.skip("the MFMA NAF kernel covers action_dim <= 2 (the generic kernel takes the rest)")
    smin, smax, amax = _bounds(S, A)
    pop = NAFPopulation(n_agents, S, A, L1, L2, B, cap, 0.01, smin, smax, amax, lr, seeds=list(range(3, 3 + n_agents)))
    if kernel != "auto":
        pop.set_kernel(kernel)
        assert pop.kernel_in_use() == kernel
    return pop


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("dims,B", CASES + [((8, 1, 128, 96), 100), ((4, 2, 256, 144), 48),
                                            ((8, 2, 200, 200), 97), ((8, 2, 200, 200), 101)])   # tail-of-four edge cases
def test_naf_hip_update_matches_oracle(hip_lib, dims, B, kernel):
    if B in (97, 101) and kernel != "mfma":
        pytest.skip("tail-of-four edge cases concern the MFMA kernel only")
    d = NafDims(*dims)
    th = init_params(d, 2)
    smin, smax, amax = _bounds(dims[0], dims[1])
    pop = _pop(dims, B, kernel=kernel)
    pop.enable_grad_taps(True)
    pop.set_params(0, th)
    o = NAFOracle(d, th, 1e-3, 0.01, smin, smax, amax)
    rng = np.random.RandomState(1)
    for it in range(3):
        s, a, s2, r, g = _batch(rng, B, dims[0], dims[1])
        pop.update_batch(0, s, a, s2, r, g)
        t = o.update(s, a, s2, r, g, taps=True)
        tol = 1e-5 if it == 0 else 2e-4
        for k in ("q", "y", "V"):
            assert _rel(pop.last_tap(0, k), t[k]) < tol, (it, k)
        if it == 0:
            got = pop.last_tap(0, "grads")
            lay, _ = d.layout()
            for n, (off, shp) in lay.items():
                k = int(np.prod(shp))
                assert _rel(got[off:off + k], t["grads"][off:off + k]) < 2e-5, n
            assert _rel(pop.get_blob(0, "theta_target"), o.theta_t) < 1e-5
            assert np.allclose(pop.get_beta_powers(0), o.pw, rtol=1e-6)
    pop.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", KERNELS)
def test_naf_hip_replay_path_act_and_agent(hip_lib, kernel):
    dims, B, N = (8, 2, 200, 200), 32, 1500
    d = NafDims(*dims)
    smin, smax, amax = _bounds(8, 2)
    pop = _pop(dims, B, n_agents=2, cap=N, kernel=kernel)
    rng = np.random.RandomState(5)
    data = (rng.uniform(-3, 3, (N, 8)), rng.uniform(-2, 2, (N, 2)), rng.uniform(-16, 0, N), rng.uniform(-3, 3, (N, 8)),
            np.full(N, 0.99))
    ths = [init_params(d, 20 + i) for i in range(2)]
    oracles = []
    for i in range(2):
        pop.set_params(i, ths[i])
        pop.replay_add_batch(i, *data)
        oracles.append(NAFOracle(d, ths[i], 1e-3, 0.01, smin, smax, amax))
    idx = np.stack([rng.choice(N, B, replace=False) for _ in range(4)]).reshape(2, 2, B).astype(np.int64)
    pop.update(2, host_indices=idx)
    for i in range(2):
        for k in range(2):
            j = idx[i, k]
            t = oracles[i].update(data[0][j], data[1][j], data[3][j], data[2][j], data[4][j], taps=True)
        for name in ("q", "y", "V"):
            assert _rel(pop.last_tap(i, name), t[name]) < 2e-4, (i, name)
    st = rng.uniform(-3, 3, (2, 8))
    mu, lc = pop.act(st, with_lcols=True)
    for i in range(2):
        wm, wl = oracles[i].act(st[i:i + 1])
        assert _rel(mu[i], wm[0]) < 1e-5 and _rel(lc[i], wl[0]) < 1e-5
    pop.update(3)                                     # device sampler path
    assert np.all(np.isfinite(pop.get_blob(1, "theta")))
    pop.close()
    # drop-in agent (Pendulum: A = 1) with the reference's host-side covariance sampling
    from rlcontrol_amd.utils.config import Config
    from rlcontrol_amd.utils.main_utils import create_agent
    from rlcontrol_amd.environments.environments import create_environment
    env = create_environment({"environment": "Pendulum-v0", "TotalMilSteps": 0.001, "EpisodeSteps": -1,
                              "EvalIntervalMilSteps": 0.0005, "EvalEpisodes": 2})
    cfg = Config()
    cfg.merge_config({"env_name": env.name, "state_dim": env.state_dim, "state_min": env.state_min,
                      "state_max": env.state_max, "action_dim": env.action_dim, "action_min": env.action_min,
                      "action_max": env.action_max, "norm_type": "input_norm", "exploration_policy": "none",
                      "l1_dim": 200, "l2_dim": 200, "noise_scale": 0.3, "learning_rate": 0.001, "buffer_size": 5000,
                      "writer": None, "write_log": False, "write_plot": False, "random_seed": 0})
    agent = create_agent("NAF", cfg)
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
    assert agent.replay_buffer.get_size() == 60
    assert np.array_equal(agent.start(obs, False), agent.start(obs, False))


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", KERNELS)
def test_naf_device_sampler_equals_oracle_on_the_same_philox_minibatches(hip_lib, kernel):
    """Fused path with the DEVICE sampler: same Philox stream as oracle/philox.py, so K updates in one launch
    must match the oracle fed with the minibatches that stream selects."""
    from oracle import philox
    dims, B, N, K = (8, 2, 200, 200), 64, 2048, 4
    d = NafDims(*dims)
    smin, smax, amax = _bounds(*dims[:2])
    pop = _pop(dims, B, cap=N, lr=1e-4, kernel=kernel)        # seeds = [3]
    th = init_params(d, 17)
    rng = np.random.RandomState(6)
    s, a, s2 = rng.uniform(-3, 3, (N, 8)), rng.uniform(-1, 1, (N, 2)), rng.uniform(-3, 3, (N, 8))
    r, g = rng.uniform(-1, 1, N), np.where(rng.rand(N) < 0.1, 0.0, 0.99)
    pop.set_params(0, th)
    pop.replay_add_batch(0, s, a, r, s2, g)
    o = NAFOracle(d, th, 1e-4, 0.01, smin, smax, amax)
    pop.update(K)
    s32, a32, s232 = s.astype(np.float32), a.astype(np.float32), s2.astype(np.float32)
    for call in range(K):
        j = philox.sample_distinct(N, B, 3, call)
        t = o.update(s32[j], a32[j], s232[j], r[j], g[j], taps=True)
    for k in ("q", "y", "V"):
        assert _rel(pop.last_tap(0, k), t[k]) < 1e-4, k
    assert _rel(pop.get_blob(0, "theta_target"), o.theta_t) < 1e-4
    pop.close()


@pytest.mark.gpu
def test_naf_kernel_switch_repacks_weights_and_optimizer_state(hip_lib):
    """generic <-> mfma changes the device layout (row-major <-> tile-blocked): blobs survive the round trip and the
    two kernels continue the same trajectory to summation-order accuracy."""
    dims, B = (8, 2, 200, 200), 100
    d = NafDims(*dims)
    th = init_params(d, 9)
    rng = np.random.RandomState(2)
    batches = [_batch(rng, B, 8, 2) for _ in range(4)]
    pa, pb = _pop(dims, B, kernel="mfma", lr=1e-4), _pop(dims, B, kernel="generic", lr=1e-4)
    for p in (pa, pb):
        p.set_params(0, th)
    for k, (s, a, s2, r, g) in enumerate(batches):
        pa.update_batch(0, s, a, s2, r, g)
        pb.update_batch(0, s, a, s2, r, g)
        if k == 1:
            before = {w: pa.get_blob(0, w) for w in ("theta", "theta_target", "adam_m", "adam_v")}
            pa.set_kernel("generic"); pb.set_kernel("mfma")
            for w, v in before.items():
                assert np.array_equal(pa.get_blob(0, w), v), w
    for w in ("theta", "theta_target", "adam_m"):
        assert _rel(pa.get_blob(0, w), pb.get_blob(0, w)) < 2e-4, w
    pa.close(); pb.close()


# ------------------------------------------------------------------------------------------ norm_type 'layer'
LN_CASES = [((3, 1, 64, 48), 17), ((8, 2, 200, 200), 32), ((5, 3, 32, 40), 9)]


def _vo(dims, th, norm_type, dtype=None):
    import torch
    from oracle.naf_variants import NafVariantOracle
    smin, smax, amax = _bounds(dims[0], dims[1])
    return NafVariantOracle(dims, th, 1e-3, 0.01, smin, smax, amax, norm_type=norm_type,
                            dtype=dtype or torch.float32)


@pytest.mark.parametrize("dims,B", LN_CASES)
def test_naf_variant_oracle_without_norm_is_the_c_oracle(dims, B):
    """the torch restatement used for layer norm, run with norm_type 'input_norm', must reproduce oracle/naf_oracle.c"""
    d = NafDims(*dims)
    th = init_params(d, 2)
    smin, smax, amax = _bounds(dims[0], dims[1])
    o = NAFOracle(d, th, 1e-3, 0.01, smin, smax, amax)
    v = _vo(dims, th, "input_norm")
    rng = np.random.RandomState(1)
    for it in range(2):
        s, a, s2, r, g = _batch(rng, B, dims[0], dims[1])
        to, tv = o.update(s, a, s2, r, g, taps=True), v.update(s, a, s2, r, g, taps=True)
        for k in ("q", "y", "V"):
            assert _rel(tv[k], to[k]) < (1e-5 if it == 0 else 2e-4), (it, k)
        if it == 0:
            lay, _ = d.layout()
            for n, (off, shp) in lay.items():
                k = int(np.prod(shp))
                assert _rel(tv["grads"][off:off + k], to["grads"][off:off + k]) < 3e-5, n
            assert _rel(v.theta_t.numpy(), o.theta_t) < 1e-5
    mu, lc = v.act(s[:4])
    wm, wl = o.act(s[:4])
    assert _rel(mu, wm) < 1e-5 and _rel(lc, wl) < 1e-5


@pytest.mark.parametrize("dims,B", LN_CASES)
def test_naf_layer_norm_oracle_agrees_with_its_float64_twin(dims, B):
    import torch
    from oracle.naf_variants import init_params as vinit, layout
    th = vinit(dims, 4, True)
    lay, P = layout(dims, True)
    assert P == NafDims(*dims).P + 2 * (dims[2] + 2 * dims[3])       # beta + gamma of the trunk and both branches
    assert list(lay)[:4] == ["W1", "b1", "L1b", "L1g"] and list(lay)[6:8] == ["La2b", "La2g"]
    rng = np.random.RandomState(3)
    th = th + (rng.uniform(-0.2, 0.2, P) * np.array([n.startswith("L") for n, (o, s) in lay.items()
                                                     for _ in range(int(np.prod(s)))])).astype(np.float32)
    o32, o64 = _vo(dims, th, "layer"), _vo(dims, th, "layer", torch.float64)
    s, a, s2, r, g = _batch(rng, B, dims[0], dims[1])
    t32, t64 = o32.update(s, a, s2, r, g, taps=True), o64.update(s, a, s2, r, g, taps=True)
    for k in ("q", "y", "V"):
        assert _rel(t32[k], t64[k]) < 1e-5, k
    for n, (off, shp) in lay.items():
        k = int(np.prod(shp))
        assert _rel(t32["grads"][off:off + k], t64["grads"][off:off + k]) < 1e-4, n
    for n in ("L1g", "L1b", "La2g", "Lv2g", "Lv2b"):                  # the layer norms are in the gradient path
        off, shp = lay[n]
        assert np.max(np.abs(t64["grads"][off:off + shp[0]])) > 0, n


def test_naf_host_layout_with_layer_norm_follows_the_oracle():
    from oracle.naf_variants import layout
    from rlcontrol_amd.hip_naf import param_layout, init_params as hinit
    for dims in ((3, 1, 64, 48), (5, 3, 32, 40)):
        for norm_type in ("input_norm", "layer"):
            lay, P = layout(dims, norm_type == "layer")
            got, gotP = param_layout(*dims, norm_type=norm_type)
            assert gotP == P and [(k, v) for k, v in got.items()] == [(k, v) for k, v in lay.items()]
        th = hinit(*dims, seed=1, norm_type="layer")
        lay, _ = layout(dims, True)
        assert np.all(th[lay["L1g"][0]:lay["L1g"][0] + dims[2]] == 1.0) and np.all(th[lay["L1b"][0]:lay["L1b"][0] + dims[2]] == 0)
    with pytest.raises(ValueError):
        param_layout(3, 1, 8, 8, norm_type="batch")


def _ln_theta(dims, seed):
    """initial blob with the layer-norm gammas / betas moved off their 1 / 0 defaults"""
    from oracle.naf_variants import init_params as vinit, layout
    lay, P = layout(dims, True)
    th = vinit(dims, seed, True)
    rng = np.random.RandomState(seed + 100)
    for n, (off, shp) in lay.items():
        if n.startswith("L"):
            th[off:off + shp[0]] += rng.uniform(-0.2, 0.2, shp[0]).astype(np.float32)
    return th, lay, P


@pytest.mark.gpu
@pytest.mark.parametrize("dims,B", LN_CASES + [((8, 1, 128, 96), 100)])
def test_naf_hip_layer_norm_update_matches_oracle(hip_lib, dims, B):
    from rlcontrol_amd.hip_naf import NAFPopulation
    th, lay, P = _ln_theta(dims, 2)
    smin, smax, amax = _bounds(dims[0], dims[1])
    pop = NAFPopulation(1, *dims, B, 512, 0.01, smin, smax, amax, 1e-3, seeds=[3], norm_type="layer")
    assert pop.P == P and pop.kernel_in_use() == "generic"
    with pytest.raises(RuntimeError):
        pop.set_kernel("mfma")
    pop.enable_grad_taps(True)
    pop.set_params(0, th)
    assert np.array_equal(pop.get_blob(0, "theta"), th) and np.array_equal(pop.get_blob(0, "theta_target"), th)
    o = _vo(dims, th, "layer")
    rng = np.random.RandomState(1)
    for it in range(3):
        s, a, s2, r, g = _batch(rng, B, dims[0], dims[1])
        pop.update_batch(0, s, a, s2, r, g)
        t = o.update(s, a, s2, r, g, taps=True)
        tol = 1e-5 if it == 0 else 3e-4
        for k in ("q", "y", "V"):
            assert _rel(pop.last_tap(0, k), t[k]) < tol, (it, k)
        if it == 0:
            got = pop.last_tap(0, "grads")
            for n, (off, shp) in lay.items():
                k = int(np.prod(shp))
                assert _rel(got[off:off + k], t["grads"][off:off + k]) < 5e-5, n
            assert _rel(pop.get_blob(0, "theta_target"), o.theta_t.numpy()) < 1e-5
    st = rng.uniform(-3, 3, (3, dims[0]))
    for i in range(3):                                               # acting path (B = 1) through the layer norms
        mu, lc = pop.act(st[i:i + 1], with_lcols=True)
        wm, wl = o.act(st[i:i + 1])
        assert _rel(mu, wm) < 3e-4 and _rel(lc, wl) < 3e-4
    pop.close()


@pytest.mark.gpu
def test_naf_hip_layer_norm_replay_path_and_batch_norm_refused(hip_lib):
    """fused sample + gather + update on the replay with layer norm, two agents, K updates in one launch"""
    from rlcontrol_amd.hip_naf import NAFPopulation
    from rlcontrol_amd import _lib
    dims, B, N = (8, 2, 64, 64), 32, 600
    smin, smax, amax = _bounds(8, 2)
    pop = NAFPopulation(2, *dims, B, N, 0.01, smin, smax, amax, 1e-3, seeds=[3, 4], norm_type="layer")
    rng = np.random.RandomState(5)
    data = (rng.uniform(-3, 3, (N, 8)), rng.uniform(-2, 2, (N, 2)), rng.uniform(-16, 0, N), rng.uniform(-3, 3, (N, 8)),
            np.full(N, 0.99))
    oracles = []
    for i in range(2):
        th, _, _ = _ln_theta(dims, 20 + i)
        pop.set_params(i, th)
        pop.replay_add_batch(i, *data)
        oracles.append(_vo(dims, th, "layer"))
    idx = np.stack([rng.choice(N, B, replace=False) for _ in range(4)]).reshape(2, 2, B).astype(np.int64)
    pop.update(2, host_indices=idx)
    for i in range(2):
        for k in range(2):
            j = idx[i, k]
            t = oracles[i].update(data[0][j], data[1][j], data[3][j], data[2][j], data[4][j], taps=True)
        for name in ("q", "y", "V"):
            assert _rel(pop.last_tap(i, name), t[name]) < 3e-4, (i, name)
    pop.update(3)                                     # device sampler path
    assert np.all(np.isfinite(pop.get_blob(1, "theta")))
    pop.close()
    # the C ABI refuses 'batch' by name (the Python layout helper refuses it before the library is reached)
    from rlcontrol_amd import hip_naf
    hip_naf.NORM_TYPES["batch"] = 2
    try:
        with pytest.raises(_lib.RlcError, match="batch"):
            NAFPopulation(1, *dims, B, N, 0.01, smin, smax, amax, 1e-3, seeds=[3], norm_type="batch")
    finally:
        del hip_naf.NORM_TYPES["batch"]
