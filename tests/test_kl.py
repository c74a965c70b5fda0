"""ReverseKL / ForwardKL: what pins the torch oracle on CPU + HIP parity on the GPU.

Parity status: "parity unpinned" against the reference itself -- its two KL network files import quadpy and gym, which
are not installed, and it ships no fixture for these agents.  The oracle (oracle/kl_torch.py) runs the reference's own
tensor library; the CPU tests below pin the pieces a restatement could get wrong: the quadrature rule (two independent
constructions + polynomial exactness), the explicit torch-1.7.1 Adam against torch.optim.Adam, the log-density against
torch.distributions.Normal and a change of variables check, and the autograd policy gradient against central differences
of the loss.  The GPU tests compare kl_generic.hip with the oracle at 1e-5 relative (fp32, summation order differs).
"""
import math

import numpy as np
import pytest
import torch

from oracle import kl_torch as K

MODES = ([("reverse", o, q) for o in K.OPTIM_TYPES for q in K.Q_UPDATE_TYPES] +
         [("forward", "intg", q) for q in K.Q_UPDATE_TYPES])
HEADLINE = (3, 1, 200, 200, 200, 200)


def _rel(x, y):
    x, y = np.asarray(x, np.float64).ravel(), np.asarray(y, np.float64).ravel()
    return float(np.max(np.abs(x - y)) / (np.max(np.abs(y)) + 1e-30))


def _batch(rng, B, S):
    return (rng.uniform(-2, 2, (B, S)), rng.uniform(-2, 2, (B, 1)), rng.uniform(-2, 2, (B, S)),
            rng.uniform(-16, 0, B), np.where(rng.rand(B) < 0.2, 0.0, 0.99), rng.randn(B, 1))


def _lively(d, th, rng):
    """the U(+-3e-3) output layers give an almost state-independent policy; widen them so every path carries signal"""
    lay, _ = d.layout()
    th = th.copy()
    for name, scale in (("pWm", 30.0), ("pWs", 30.0), ("qW3", 30.0), ("vW3", 30.0)):
        off, shp = lay[name]
        th[off:off + int(np.prod(shp))] *= scale
    return th


# ----------------------------------------------------------------------------------------- CPU: quadrature
@pytest.mark.parametrize("n", [3, 4, 9, 17, 64, 65, 254])
def test_clenshaw_curtis_two_constructions_and_exactness(n):
    from rlcontrol_amd.utils.quadrature import clenshaw_curtis
    x, w = clenshaw_curtis(n)
    x2, w2 = K.cc_rule(n)
    assert np.max(np.abs(x - x2)) < 1e-15 and np.max(np.abs(w - w2)) < 1e-15
    assert np.all(np.diff(x) > 0) and x[0] == -1.0 and x[-1] == 1.0 and np.all(w > 0)
    for deg in range(n):                      # an n-point interpolatory rule integrates degree n-1 exactly
        exact = 0.0 if deg % 2 else 2.0 / (deg + 1)
        assert abs(np.sum(w * x ** deg) - exact) < 1e-13, deg


def test_interior_nodes_match_the_oracle_fp32_values():
    from rlcontrol_amd.utils.quadrature import interior_action_nodes
    a, w = interior_action_nodes(64, 2.0)
    o = K.KLOracle("reverse", K.KlDims(*HEADLINE), K.init_params(K.KlDims(*HEADLINE), 0), 1e-3, 1e-3, 0.1, 0.01, 2.0, 64)
    assert a.dtype == np.float32 and len(a) == 62
    assert np.array_equal(a, o.nodes.numpy().reshape(-1)) and np.array_equal(w, o.weights.numpy())
    assert np.all(np.abs(a) < 2.0)
    # the Gaussian policy's squashed density integrates to 1 under the rule when it is not too peaked
    mean, std = 0.3, 0.8
    u = np.arctanh(a.astype(np.float64) / 2.0)
    dens = np.exp(-(u - mean) ** 2 / (2 * std ** 2)) / (std * math.sqrt(2 * math.pi)) / (1 - (a / 2.0) ** 2)
    assert abs(np.sum(w * dens) - 1.0) < 1e-3


# ----------------------------------------------------------------------------------------- action_dim > 1
@pytest.mark.parametrize("l,A", [(5, 2), (6, 2), (5, 3)])
def test_sparse_grid_two_constructions_and_exactness(l, A):
    """the product's sparse grid (closed-form Clenshaw-Curtis) equals the oracle's restatement of
    reversekl_network.py:78-108 (FFT Clenshaw-Curtis, the reference's loop); with the end points kept it is Smolyak's
    rule: exact on every monomial of total degree <= 2 (l - A) + 1 at least"""
    from rlcontrol_amd.utils.quadrature import sparse_grid_action_nodes
    amax = np.linspace(1.0, 2.0, A)
    a, w = sparse_grid_action_nodes(l, A, amax)
    oa, ow = K.sparse_grid(l, A, amax)
    assert a.shape == oa.shape == (len(w), A) and a.dtype == np.float32
    assert np.array_equal(a, oa) and np.allclose(w, ow, rtol=2e-6, atol=1e-7)
    assert np.all(np.abs(a) < amax[None, :]) and (w < 0).any()          # interior nodes; combination weights of both signs
    af, wf = sparse_grid_action_nodes(l, A, 1.0, interior=False)
    af, wf = af.astype(np.float64), wf.astype(np.float64)
    import itertools
    for deg in itertools.product(range(4), repeat=A):
        if sum(deg) > 2 * (l - A) + 1:
            continue
        exact = np.prod([0.0 if p % 2 else 2.0 / (p + 1) for p in deg])
        assert abs(np.sum(wf * np.prod(af ** np.array(deg)[None, :], 1)) - exact) < 2e-5, deg


def test_multivariate_policy_keeps_the_reference_covariance():
    """get_distribution: MultivariateNormal(mean, diag_embed(std)) -- the covariance is diag(std), so component j has
    variance std_j; the oracle's draw and log-density follow that class, not N(mean, std^2)"""
    mean, std = torch.tensor([[0.2, -0.4]]), torch.tensor([[0.25, 1.5]])
    z = torch.tensor([[0.7, 0.1]])
    lp = K._normal_logprob(z, mean, std)
    want = sum(-(z[0, j] - mean[0, j]) ** 2 / (2 * std[0, j]) - 0.5 * torch.log(std[0, j]) - 0.5 * math.log(2 * math.pi)
               for j in range(2))
    assert lp.shape == (1, 1) and abs(lp.item() - want.item()) < 1e-6
    assert torch.allclose(K._sample_scale(std), std.sqrt()) and torch.equal(K._sample_scale(std[:, :1]), std[:, :1])
    torch.manual_seed(0)
    draws = torch.distributions.MultivariateNormal(mean[0], torch.diag_embed(std[0])).sample((20000,))
    assert torch.allclose(draws.var(0), std[0], rtol=0.05)


MULTI = [((3, 2, 48, 40, 44, 36), 12, 5), ((4, 3, 32, 32, 32, 32), 9, 4), ((3, 2, 64, 64, 64, 64), 32, 6)]


def _batch_a(rng, B, S, A):
    return (rng.uniform(-2, 2, (B, S)), rng.uniform(-2, 2, (B, A)), rng.uniform(-2, 2, (B, S)),
            rng.uniform(-16, 0, B), np.where(rng.rand(B) < 0.2, 0.0, 0.99), rng.randn(B, A))


@pytest.mark.gpu
@pytest.mark.parametrize("kind,optim,qup", MODES)
@pytest.mark.parametrize("dims,B,l_param", MULTI)
def test_kl_hip_multi_dimensional_actions_match_oracle(hip_lib, kind, optim, qup, dims, B, l_param):
    """action_dim > 1: sparse-grid nodes, the diag(std)-covariance policy, A action rows at the Q network's input"""
    d = K.KlDims(*dims)
    S, A = dims[0], dims[1]
    rng = np.random.RandomState(3)
    th = _lively(d, K.init_params(d, 1), rng)
    amax = np.full(A, 2.0)
    pop = _pop(kind, dims, B, optim, qup, l_param=l_param, action_max=amax)
    assert pop.kernel_in_use() == "generic"
    pop.enable_grad_taps(True)
    pop.set_params(0, th)
    o = K.KLOracle(kind, d, th, 1e-3, 1e-2, 0.3, 0.01, 2.0, 0, optim, qup, l_param=l_param, action_max=amax)
    assert pop.n_nodes == len(o.weights)
    lay, _ = d.layout()
    st, e1 = rng.uniform(-2, 2, (4, S)), rng.randn(4, A)
    for i in range(4):                                   # acting: mean action and the sqrt(std)-scaled sample
        assert _rel(pop.act(st[i:i + 1]), o.act(st[i:i + 1])) < 1e-5
        assert _rel(pop.act(st[i:i + 1], sample=True, eps=e1[i:i + 1]), o.act(st[i:i + 1], eps=e1[i:i + 1])) < 1e-5
    for it in range(3):
        s, a, s2, r, g, eps = _batch_a(rng, B, S, A)
        pop.update_batch(0, s, a, s2, r, g, eps=eps)
        t = o.update(s, a, s2, r, g, eps, taps=True)
        tol = 1e-5 if it == 0 else 2e-4
        for k in ("q", "v", "q_pi", "logp"):
            assert _rel(pop.last_tap(0, k), t[k]) < tol, (it, k)
        if "intgrl_q" in t:
            assert _rel(pop.last_tap(0, "intgrl_q"), t["intgrl_q"]) < tol, it
        assert _rel(pop.last_tap(0, "loss"), t["loss"]) < 10 * tol, it
        if it == 0:
            got = pop.last_tap(0, "grads")
            for n, (off, shp) in lay.items():
                k = int(np.prod(shp))
                # the sparse grid's weights cancel (both signs): the policy gradient is a difference of large sums
                bound = 3e-4 if (n[0] == "p" or k == 1) else 5e-5
                assert _rel(got[off:off + k], t["grads"][off:off + k]) < bound, n
            vo = lay["vW1"][0]
            assert _rel(pop.get_blob(0, "theta_target")[vo:], o.theta_t.numpy()[vo:]) < 1e-5
    pop.close()


# ----------------------------------------------------------------------------------------- CPU: oracle pieces
def test_explicit_adam_171_equals_torch_optim_adam():
    rng = np.random.RandomState(0)
    p0 = torch.tensor(rng.randn(500).astype(np.float32))
    pa = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pa], lr=3e-3)
    pb, m, v = p0.clone(), torch.zeros(500), torch.zeros(500)
    for step in range(1, 40):
        g = torch.tensor((rng.randn(500) * 10.0 ** rng.uniform(-4, 1)).astype(np.float32))
        pa.grad = g.clone()
        opt.step()
        K.adam_171(pb, g, m, v, step, 3e-3)
        assert _rel(pb.numpy(), pa.detach().numpy()) < 1e-6, step


def test_log_density_is_torch_normal_and_changes_variables():
    mean, std = torch.tensor([[0.2], [-1.0]]), torch.tensor([[0.5], [1.5]])
    val = torch.tensor([[0.7], [0.1]])
    assert torch.allclose(K._normal_logprob(val, mean, std), torch.distributions.Normal(mean, std).log_prob(val), atol=1e-7)


@pytest.mark.parametrize("kind,optim,qup", MODES)
def test_oracle_policy_gradient_matches_central_differences(kind, optim, qup):
    """autograd's d policy_loss / d (mean bias, log_std bias) against a float64 central difference of the loss the
    reference writes down, evaluated by an independent numpy forward pass"""
    dims, B = (3, 1, 24, 20, 28, 16), 6
    d = K.KlDims(*dims)
    rng = np.random.RandomState(4)
    th = _lively(d, K.init_params(d, 2), rng)
    s, a, s2, r, g, eps = _batch(rng, B, 3)
    alpha, amax = 0.3, 2.0
    o = K.KLOracle(kind, d, th, 1e-3, 1e-3, alpha, 0.01, amax, 16, optim, qup)
    nodes, wts = o.nodes.numpy().reshape(-1).astype(np.float64), o.weights.numpy().astype(np.float64)
    t = o.update(s, a, s2, r, g, eps, taps=True)
    lay, _ = d.layout()

    def np_loss(theta):
        p = {k: theta[off:off + int(np.prod(shp))].reshape(shp).astype(np.float64) for k, (off, shp) in lay.items()}
        relu = lambda x: np.maximum(x, 0.0)
        mlp = lambda pre, x: relu(relu(x @ p[pre + "W1"] + p[pre + "b1"]) @ p[pre + "W2"] + p[pre + "b2"]) @ p[pre + "W3"] + p[pre + "b3"]
        h = relu(relu(s @ p["pW1"] + p["pb1"]) @ p["pW2"] + p["pb2"])
        mean, ls = h @ p["pWm"] + p["pbm"], np.clip(h @ p["pWs"] + p["pbs"], -20, 2)
        std = np.exp(ls)
        logn = lambda x: -(x - mean) ** 2 / (2 * std ** 2) - ls - 0.5 * math.log(2 * math.pi)
        v = mlp("v", s)
        if optim in ("ll", "hard_ll"):
            z = t["z"].reshape(B, 1)
            lp = logn(z) - np.log(1 - np.tanh(z) ** 2 + 1e-6)
            adv = t["q_pi"].reshape(B, 1) - t["v"].reshape(B, 1) - (alpha * t["logp"].reshape(B, 1) if optim == "ll" else 0.0)
            return float(np.mean(-lp * adv))
        x = nodes / amax
        lp = logn(np.arctanh(x)[None, :]) - np.log(1 - x ** 2 + 1e-6)[None, :]          # [B, K]
        iq = t["intgrl_q"].reshape(B, -1).astype(np.float64)
        if kind == "reverse":
            adv = iq - t["v"].reshape(B, 1)
            f = -np.exp(lp) * (adv - alpha * lp if optim == "intg" else adv)
            return float(np.mean((f * wts).sum(-1)))
        sc = iq / alpha
        e = np.exp(sc - sc.max(-1, keepdims=True))
        bp = e / (e * wts).sum(-1, keepdims=True)
        return float(np.mean(-(bp * lp * wts).sum(-1)))

    th64 = th.astype(np.float64)
    for name in ("pbm", "pbs"):
        off = lay[name][0]
        hstep = 1e-5
        up, dn = th64.copy(), th64.copy()
        up[off] += hstep
        dn[off] -= hstep
        fd = (np_loss(up) - np_loss(dn)) / (2 * hstep)
        assert abs(fd - t["grads"][off]) < 2e-3 * max(1.0, abs(fd)), (name, fd, t["grads"][off])


def test_oracle_targets_and_polyak_follow_the_reference_lines():
    dims, B = (3, 1, 16, 16, 16, 16), 8
    d = K.KlDims(*dims)
    rng = np.random.RandomState(1)
    th = _lively(d, K.init_params(d, 3), rng)
    s, a, s2, r, g, eps = _batch(rng, B, 3)
    lay, P = d.layout()
    vo = lay["vW1"][0]
    for qup in K.Q_UPDATE_TYPES:
        o = K.KLOracle("reverse", d, th, 1e-3, 1e-2, 0.2, 0.05, 2.0, 8, "intg", qup)
        tgt0 = o.theta_t.numpy().copy()
        t = o.update(s, a, s2, r, g, eps, taps=True)
        # q regresses onto r + gamma * V'(s'); V onto (r - alpha logp) + gamma V' (non_sac) or Q(s, a_new) - alpha logp (sac)
        v_next = (t["q_target"] - r.astype(np.float32)) / np.where(g > 0, g, 1.0)
        want_v = (r - 0.2 * t["logp"]) + g * v_next if qup == "non_sac" else t["q_pi"] - 0.2 * t["logp"]
        assert abs(t["loss"][2] - np.mean((t["v"] - want_v) ** 2)) < 1e-4 * max(1.0, t["loss"][2])
        assert abs(t["loss"][1] - np.mean((t["q"] - t["q_target"]) ** 2)) < 1e-5 * max(1.0, t["loss"][1])
        # target: only the V block moves, by tau
        new_t = o.theta_t.numpy()
        assert np.array_equal(new_t[:vo], tgt0[:vo])
        assert _rel(new_t[vo:], tgt0[vo:] * 0.95 + o.theta.numpy()[vo:] * 0.05) < 1e-6
        assert o.step == 1


# ------------------------------------------------------------------------------------------ GPU
KERNELS = ["generic", "mfma"]


def _pop(kind, dims, B, optim="intg", qup="non_sac", n_agents=1, alpha=0.3, cap=2048, n_param=64, pi_lr=1e-3, qv_lr=1e-2,
         kernel="auto", l_param=None, action_max=None):
    from rlcontrol_amd.hip_kl import KLPopulation
    from rlcontrol_amd._lib import RlcError
    S, A, L1A, L2A, L1C, L2C = dims
    pop = KLPopulation(kind, n_agents, S, A, L1A, L2A, L1C, L2C, B, cap, 0.01, 2.0, pi_lr, qv_lr, alpha,
                       seeds=list(range(5, 5 + n_agents)), n_param=n_param, optim_type=optim, q_update_type=qup,
                       l_param=l_param, action_max=action_max)
    if kernel != "auto":
        try:
            pop.set_kernel(kernel)
        except RlcError:
            pop.close()
            pytest.skip("the %s kernel does not take this shape" % kernel)
        assert pop.kernel_in_use() == kernel
    return pop


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("kind,optim,qup", MODES)
@pytest.mark.parametrize("dims,B,n_param", [(HEADLINE, 32, 64), ((5, 1, 64, 48, 40, 56), 17, 9), ((3, 1, 128, 128, 128, 128), 100, 33),
                                            ((7, 1, 200, 160, 176, 144), 24, 130)])
def test_kl_hip_update_matches_oracle(hip_lib, kind, optim, qup, dims, B, n_param, kernel):
    d = K.KlDims(*dims)
    rng = np.random.RandomState(3)
    th = _lively(d, K.init_params(d, 1), rng)
    pop = _pop(kind, dims, B, optim, qup, n_param=n_param, kernel=kernel)
    pop.enable_grad_taps(True)
    pop.set_params(0, th)
    o = K.KLOracle(kind, d, th, 1e-3, 1e-2, 0.3, 0.01, 2.0, n_param, optim, qup)
    lay, _ = d.layout()
    for it in range(3):
        s, a, s2, r, g, eps = _batch(rng, B, dims[0])
        pop.update_batch(0, s, a, s2, r, g, eps=eps)
        t = o.update(s, a, s2, r, g, eps, taps=True)
        tol = 1e-5 if it == 0 else 2e-4         # later updates compare trajectories (Adam amplifies rounding)
        for k in ("q", "v", "q_pi", "logp"):
            assert _rel(pop.last_tap(0, k), t[k]) < tol, (it, k)
        if "intgrl_q" in t:
            assert _rel(pop.last_tap(0, "intgrl_q"), t["intgrl_q"]) < tol, it
        assert _rel(pop.last_tap(0, "loss"), t["loss"]) < 10 * tol, it
        if it == 0:
            got = pop.last_tap(0, "grads")
            for n, (off, shp) in lay.items():
                k = int(np.prod(shp))
                # a one-element tensor (a head's bias) is a cancelling sum over the batch measured against itself
                assert _rel(got[off:off + k], t["grads"][off:off + k]) < (5e-5 if k > 1 else 3e-4), n
            # the first Adam step is lr * g / (|g| + eps): where |g| is a cancelling sum near zero, rounding turns into a
            # step of up to lr -- parameters are compared where the step is well conditioned, and bounded elsewhere
            got_th, want_th = pop.get_blob(0, "theta"), o.theta.numpy()
            for n, (off, shp) in lay.items():
                k = int(np.prod(shp))
                lr = 1e-3 if n[0] == "p" else 1e-2
                gref = np.abs(t["grads"][off:off + k])
                solid = gref > 1e-3 * gref.max()
                dth = np.abs(got_th[off:off + k].astype(np.float64) - want_th[off:off + k])
                assert dth[solid].max() < 2e-3 * lr + 1e-7, n
                assert dth.max() <= 1.05 * lr, n
            assert _rel(pop.get_blob(0, "adam_m"), o.m.numpy()) < 5e-5
            assert _rel(pop.get_blob(0, "adam_v"), o.v.numpy()) < 5e-5
            vo = lay["vW1"][0]
            assert _rel(pop.get_blob(0, "theta_target")[vo:], o.theta_t.numpy()[vo:]) < 1e-5
    assert pop.get_step(0) == 3
    diff = np.abs(pop.get_blob(0, "theta").astype(np.float64) - o.theta.numpy())
    assert np.quantile(diff, 0.995) < 5e-5 and diff.max() <= 3.2e-2, (np.quantile(diff, 0.995), diff.max())
    pop.close()


def _near_relu_kink(o, s, a, margin=1.5e-6):
    """True when a hidden pre-activation of a network that is differentiated on this minibatch (pi, Q(s,a), V) lies
    within fp32 rounding of zero: the two implementations may then disagree on that unit's ReLU mask, its weight
    gradients change discretely and Adam turns that into a step of order lr -- a property of the problem, not of either
    implementation, so a trajectory comparison must not feed such a minibatch."""
    p = K._views(o.theta, o.lay)
    s = torch.tensor(np.asarray(s, np.float32))
    xq = torch.cat([s, torch.tensor(np.asarray(a, np.float32))], 1)
    worst = float("inf")
    for pre, x in (("p", s), ("q", xq), ("v", s)):
        z1 = x @ p[pre + "W1"] + p[pre + "b1"]
        z2 = torch.relu(z1) @ p[pre + "W2"] + p[pre + "b2"]
        worst = min(worst, z1.abs().min().item(), z2.abs().min().item())
    return worst < margin


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("kind", K.KINDS)
def test_kl_hip_at_reference_initialisation(hip_lib, kind, kernel):
    """the reference's own initialisation (output layers U(+-3e-3)) and its json's learning rates, a trajectory of
    fifteen updates (minibatches that sit on a ReLU kink are skipped, see _near_relu_kink)"""
    dims, B = HEADLINE, 32
    d = K.KlDims(*dims)
    th = K.init_params(d, 4)
    pop = _pop(kind, dims, B, alpha=0.1, pi_lr=1e-3, qv_lr=1e-3, kernel=kernel)
    pop.set_params(0, th)
    o = K.KLOracle(kind, d, th, 1e-3, 1e-3, 0.1, 0.01, 2.0, 64)
    rng = np.random.RandomState(9)
    done = 0
    while done < 15:
        s, a, s2, r, g, eps = _batch(rng, B, 3)
        if _near_relu_kink(o, s, a):
            continue
        pop.update_batch(0, s, a, s2, r, g, eps=eps)
        t = o.update(s, a, s2, r, g, eps, taps=True)
        done += 1
    for k in ("q", "v", "q_pi", "logp", "intgrl_q"):
        assert _rel(pop.last_tap(0, k), t[k]) < 1e-4, k
    diff = np.abs(pop.get_blob(0, "theta").astype(np.float64) - o.theta.numpy())
    assert diff.max() < 2e-5 * done, (done, diff.max())          # lr = 1e-3: far below one Adam step
    pop.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("kind", K.KINDS)
def test_kl_hip_replay_path_device_sampler_and_act(hip_lib, kind, kernel):
    from oracle import philox
    from oracle.cpu_baseline import synthetic_pendulum_replay
    dims, B, N, NU = HEADLINE, 32, 2048, 3
    d = K.KlDims(*dims)
    rng = np.random.RandomState(1)
    pop = _pop(kind, dims, B, n_agents=2, cap=N, kernel=kernel)            # seeds 5, 6
    ths = [_lively(d, K.init_params(d, 10 + i), rng) for i in range(2)]
    s, a, r, s2, g = synthetic_pendulum_replay(N, 0)
    for i in range(2):
        pop.set_params(i, ths[i])
        pop.replay_add_batch(i, s, a, r, s2, g)
    oracles = [K.KLOracle(kind, d, ths[i], 1e-3, 1e-2, 0.3, 0.01, 2.0, 64) for i in range(2)]
    # host indices
    idx = np.stack([rng.choice(N, B, replace=False) for _ in range(4)]).reshape(2, 2, B).astype(np.int64)
    eps = rng.randn(2, 2, B, 1)
    pop.update(2, host_indices=idx, eps=eps)
    for i in range(2):
        for k in range(2):
            j = idx[i, k]
            t = oracles[i].update(s[j], a[j], s2[j], r[j], g[j], eps[i, k], taps=True)
        for name in ("q", "v", "q_pi", "intgrl_q"):
            assert _rel(pop.last_tap(i, name), t[name]) < 2e-4, (i, name)
    # device sampler: the Philox stream oracle/philox.py restates bit for bit
    eps = rng.randn(2, NU, B, 1)
    pop.update(NU, eps=eps)
    for i in range(2):
        for call in range(NU):
            j = philox.sample_distinct(N, B, 5 + i, call)
            t = oracles[i].update(s[j], a[j], s2[j], r[j], g[j], eps[i, call], taps=True)
        for name in ("q", "v", "q_pi"):
            assert _rel(pop.last_tap(i, name), t[name]) < 5e-4, (i, name)
        assert pop.get_step(i) == 2 + NU
    # acting
    st = rng.uniform(-2, 2, (2, 3))
    assert _rel(pop.act(st), np.stack([oracles[i].act(st[i:i + 1])[0] for i in range(2)])) < 1e-4
    e = rng.randn(2, 1)
    want = np.stack([oracles[i].act(st[i:i + 1], eps=e[i:i + 1])[0] for i in range(2)])
    assert _rel(pop.act(st, sample=True, eps=e), want) < 1e-4
    draws = np.array([pop.act(st, sample=True)[:, 0] for _ in range(200)])
    assert np.all(np.isfinite(draws)) and np.all(np.abs(draws) <= 2.0) and draws.std(0).min() > 1e-4
    pop.update(2)                                            # device sampler + device eps
    assert np.all(np.isfinite(pop.get_blob(0, "theta")))
    pop.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", K.KINDS)
@pytest.mark.parametrize("C,n_agents,n_param", [(2, 3, 64), (4, 1, 64), (8, 2, 64), (8, 9, 130), (7, 1, 9)])
def test_kl_latency_mode_equals_one_workgroup(hip_lib, kind, C, n_agents, n_param):
    """rlc_kl_set_split deals the node passes of the action integral over C workgroups per agent; the arithmetic of a pass
    does not depend on who runs it, so Q at the nodes equals the one-workgroup kernel's BIT FOR BIT and everything
    downstream to the last bit or two (the same source in a second kernel instantiation: the compiler's fma contraction
    choices may differ) -- host indices, device sampler, staged minibatch; more agents than XCDs; fewer passes than
    workgroups"""
    from oracle.cpu_baseline import synthetic_pendulum_replay
    from rlcontrol_amd._lib import RlcError
    dims, B, N = HEADLINE, 32, 1024
    d = K.KlDims(*dims)
    rng = np.random.RandomState(7)
    s, a, r, s2, g = synthetic_pendulum_replay(N, 0)
    pops = []
    for split in (1, C):
        pop = _pop(kind, dims, B, n_agents=n_agents, cap=N, n_param=n_param, kernel="mfma")
        for i in range(n_agents):
            pop.set_params(i, _lively(d, K.init_params(d, 10 + i), np.random.RandomState(i)))
            pop.replay_add_batch(i, s, a, r, s2, g)
        pop.set_split(split)
        pops.append(pop)
    idx = np.stack([rng.choice(N, B, replace=False) for _ in range(3 * n_agents)]).reshape(n_agents, 3, B).astype(np.int64)
    eps = rng.randn(n_agents, 3, B, 1)
    mb = (rng.uniform(-2, 2, (B, 3)), rng.uniform(-2, 2, (B, 1)), rng.uniform(-2, 2, (B, 3)), rng.uniform(-16, 0, B),
          np.full(B, 0.99), rng.randn(B, 1))
    for pop in pops:
        pop.update(3, host_indices=idx, eps=eps)          # K updates in one launch: 2 barriers each
        pop.update(2)                                     # device sampler + device eps
        pop.update_batch(n_agents - 1, *mb[:5], eps=mb[5])
    one, many = pops
    for i in range(n_agents):
        for blob in ("theta", "theta_target", "adam_m", "adam_v"):
            x, y = one.get_blob(i, blob).astype(np.float64), many.get_blob(i, blob).astype(np.float64)
            assert np.max(np.abs(x - y)) <= 3e-5 * np.max(np.abs(x)) + 1e-9, (i, blob)     # six Adam steps amplify a last-bit difference
        assert one.get_step(i) == many.get_step(i)
    for name in ("q", "v", "q_pi", "logp", "intgrl_q"):
        assert _rel(many.last_tap(n_agents - 1, name), one.last_tap(n_agents - 1, name)) < 1e-5, name
    assert _rel(many.last_tap(n_agents - 1, "loss"), one.last_tap(n_agents - 1, "loss")) < 1e-5
    # one update from identical parameters: the integral itself is bit-identical
    th = [one.get_blob(i, "theta") for i in range(n_agents)]
    for pop in pops:
        for i in range(n_agents):
            pop.set_params(i, th[i])
        pop.update(1, host_indices=idx[:, :1], eps=eps[:, :1])
    for i in range(n_agents):
        assert np.array_equal(one.last_tap(i, "intgrl_q"), many.last_tap(i, "intgrl_q")), i
    many.set_split(1)                                     # back to one workgroup on the same handle: the same kernel again
    for pop in pops:
        for i in range(n_agents):
            pop.set_params(i, th[i])
            pop.set_blob(i, "adam_m", np.zeros_like(th[i])); pop.set_blob(i, "adam_v", np.zeros_like(th[i]))
            pop.set_step(i, 0)
        pop.update(1, host_indices=idx[:, :1], eps=eps[:, :1])
    assert np.array_equal(one.get_blob(0, "theta"), many.get_blob(0, "theta"))
    for pop in pops:
        pop.close()


@pytest.mark.gpu
def test_kl_latency_mode_refusals(hip_lib):
    from rlcontrol_amd._lib import RlcError
    pop = _pop("reverse", HEADLINE, 32, optim="ll", kernel="mfma")
    with pytest.raises(RlcError, match="integral"):
        pop.set_split(4)
    pop.close()
    pop = _pop("reverse", HEADLINE, 32, kernel="generic")
    with pytest.raises(RlcError, match="MFMA"):
        pop.set_split(4)
    pop.close()
    pop = _pop("forward", HEADLINE, 32, n_agents=40, kernel="mfma")
    with pytest.raises(RlcError, match="co-resident"):
        pop.set_split(8)                                   # 40 agents x 8 workgroups > 256 CUs
    with pytest.raises(RlcError):
        pop.set_split(9)
    pop.set_split(4)
    pop.set_kernel("generic")                              # leaving the MFMA kernel leaves latency mode
    pop.update_batch(0, np.zeros((32, 3)), np.zeros((32, 1)), np.zeros((32, 3)), np.zeros(32), np.ones(32), eps=np.zeros((32, 1)))
    assert np.all(np.isfinite(pop.get_blob(0, "theta")))
    pop.close()


@pytest.mark.gpu
def test_kl_hip_refuses_what_it_does_not_implement(hip_lib):
    from rlcontrol_amd._lib import RlcError
    with pytest.raises(ValueError, match="l_param"):            # above one action dimension the sparse grid needs its level
        _pop("reverse", (3, 2, 32, 32, 32, 32), 8)
    with pytest.raises(RlcError, match="action_dim"):
        _pop("reverse", (3, 7, 32, 32, 32, 32), 8, l_param=7)
    with pytest.raises(RlcError, match="intg"):
        _pop("forward", (3, 1, 32, 32, 32, 32), 8, optim="ll")
    with pytest.raises(ValueError):
        _pop("reverse", (3, 1, 32, 32, 32, 32), 8, qup="td")
    pop = _pop("reverse", (3, 1, 32, 32, 32, 32), 8)
    with pytest.raises(RlcError, match="SoftActorCritic"):     # a KL handle is not an SAC handle
        from rlcontrol_amd._lib import check
        import ctypes
        out = ctypes.c_int64(0)
        check(pop._lib.rlc_sac_param_count(pop._h, ctypes.byref(out)))
    pop.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ReverseKL", "ForwardKL"])
def test_kl_dropin_agent_runs_on_pendulum(hip_lib, name):
    from rlcontrol_amd.utils.config import Config
    from rlcontrol_amd.utils.main_utils import create_agent
    from rlcontrol_amd.environments.environments import create_environment
    env = create_environment({"environment": "Pendulum-v0", "TotalMilSteps": 0.001, "EpisodeSteps": -1,
                              "EvalIntervalMilSteps": 0.0005, "EvalEpisodes": 2})
    cfg = Config()
    cfg.merge_config({"env_name": env.name, "state_dim": env.state_dim, "state_min": env.state_min,
                      "state_max": env.state_max, "action_dim": env.action_dim, "action_min": env.action_min,
                      "action_max": env.action_max})
    cfg.merge_config({"norm_type": "input_norm", "exploration_policy": "none", "actor_l1_dim": 200,
                      "actor_l2_dim": 200, "critic_l1_dim": 200, "critic_l2_dim": 200, "pi_lr": 1e-3,
                      "qf_vf_lr": 1e-3, "sample_for_eval": "False", "use_true_q": "False", "entropy_scale": 0.1,
                      "l_param": 6, "N_param": 64, "optim_type": "intg", "q_update_type": "non_sac",
                      "buffer_size": 5000, "writer": None, "write_log": False, "write_plot": False, "random_seed": 0})
    agent = create_agent(name, cfg)
    env.set_random_seed(0)
    obs = env.reset()
    agent.reset()
    a = agent.start(obs, True)
    for t in range(80):
        obs_n, r, done, _ = env.step(a)
        agent.update(obs, obs_n, float(r), a, done, False)
        a = agent.step(obs_n, True)
        obs = obs_n
        assert a.shape == (1,) and abs(a[0]) <= 2.0
    assert agent.replay_buffer.get_size() == 80
    assert agent.network_manager.population.get_step(0) == 80 - 32        # learn() once the buffer exceeds the batch
    g1, g2 = agent.start(obs, False), agent.start(obs, False)
    assert np.array_equal(g1, g2)                    # evaluation uses the mean action


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ReverseKL", "ForwardKL"])
def test_kl_dropin_agent_takes_a_two_dimensional_action_box(hip_lib, name):
    """action_dim 2 (no such environment can be built here: gym is absent): the agent builds the l_param sparse grid,
    acts, stores and learns on synthetic transitions"""
    from rlcontrol_amd.utils.config import Config
    from rlcontrol_amd.utils.main_utils import create_agent
    cfg = Config()
    cfg.merge_config({"env_name": "synthetic", "state_dim": 4, "state_min": -np.ones(4), "state_max": np.ones(4),
                      "action_dim": 2, "action_min": -np.ones(2), "action_max": np.ones(2)})
    cfg.merge_config({"norm_type": "input_norm", "exploration_policy": "none", "actor_l1_dim": 32, "actor_l2_dim": 32,
                      "critic_l1_dim": 32, "critic_l2_dim": 32, "pi_lr": 1e-3, "qf_vf_lr": 1e-3,
                      "sample_for_eval": "False", "use_true_q": "False", "entropy_scale": 0.1, "l_param": 5, "N_param": 64,
                      "optim_type": "intg", "q_update_type": "non_sac", "buffer_size": 500, "writer": None,
                      "write_log": False, "write_plot": False, "random_seed": 0})
    agent = create_agent(name, cfg)
    assert agent.network_manager.population.n_nodes == 73                  # level 5 in two dimensions
    rng = np.random.RandomState(0)
    obs = rng.uniform(-1, 1, 4)
    agent.reset()
    a = agent.start(obs, True)
    for t in range(50):
        obs_n = rng.uniform(-1, 1, 4)
        agent.update(obs, obs_n, float(-np.sum(a ** 2)), a, False, False)
        a = agent.step(obs_n, True)
        obs = obs_n
        assert a.shape == (2,) and np.all(np.abs(a) <= 1.0)
    assert agent.network_manager.population.get_step(0) == 50 - 32
    assert np.all(np.isfinite(agent.network_manager.population.get_blob(0, "theta")))


@pytest.mark.gpu
def test_kl_kernel_switch_repacks_weights_and_optimizer_state(hip_lib):
    """generic <-> mfma changes the device layout (row-major <-> tile-blocked): blobs survive the round trip and the
    two kernels continue the same trajectory to summation-order accuracy."""
    dims, B = HEADLINE, 32
    d = K.KlDims(*dims)
    rng = np.random.RandomState(5)
    th = _lively(d, K.init_params(d, 7), rng)
    o = K.KLOracle("reverse", d, th, 1e-3, 1e-2, 0.3, 0.01, 2.0, 64)
    pa, pb = _pop("reverse", dims, B, kernel="mfma"), _pop("reverse", dims, B, kernel="generic")
    for p in (pa, pb):
        p.set_params(0, th)
    done = 0
    while done < 4:
        s, a, s2, r, g, eps = _batch(rng, B, 3)
        if _near_relu_kink(o, s, a):
            continue
        o.update(s, a, s2, r, g, eps)
        pa.update_batch(0, s, a, s2, r, g, eps=eps)
        pb.update_batch(0, s, a, s2, r, g, eps=eps)
        done += 1
        if done == 2:                                # swap the kernels mid-trajectory
            before = {w: pa.get_blob(0, w) for w in ("theta", "adam_m", "adam_v")}
            pa.set_kernel("generic"); pb.set_kernel("mfma")
            for w, v in before.items():
                assert np.array_equal(pa.get_blob(0, w), v), w
    lay, _ = d.layout()
    vo = lay["vW1"][0]
    for w in ("theta", "adam_m", "adam_v"):
        assert _rel(pa.get_blob(0, w), pb.get_blob(0, w)) < 2e-4, w
    assert _rel(pa.get_blob(0, "theta_target")[vo:], pb.get_blob(0, "theta_target")[vo:]) < 2e-4
    assert _rel(pa.get_blob(0, "theta"), o.theta.numpy()) < 2e-4
    pa.close(); pb.close()


# ----------------------------------------------------------------------------------------- CPU: host mirror
def test_product_layout_and_initialiser_families_match_the_oracle():
    """rlcontrol_amd.hip_kl (product) and oracle/kl_torch.py (checker) restate the blob layout and the initialiser
    families independently; they must agree (same RandomState stream -> same numbers)."""
    from rlcontrol_amd import hip_kl
    for dims in (HEADLINE, (7, 1, 24, 20, 28, 16)):
        lay_p, P_p = hip_kl.param_layout(*dims)
        lay_o, P_o = K.KlDims(*dims).layout()
        assert P_p == P_o and list(lay_p.items()) == list(lay_o.items())
        assert np.array_equal(hip_kl.init_params(*dims, 3), K.init_params(K.KlDims(*dims), 3))
    th = hip_kl.init_params(*HEADLINE, 0)
    lay, _ = hip_kl.param_layout(*HEADLINE)
    for name, (off, shp) in lay.items():
        blk = th[off:off + int(np.prod(shp))]
        lim = 3e-3 if name[2:] in ("m", "s", "3") else 1.0 / math.sqrt({"p": {"1": 3, "2": 200}, "q": {"1": 4, "2": 200},
                                                                      "v": {"1": 3, "2": 200}}[name[0]][name[2:]])
        assert np.abs(blk).max() <= lim and (blk.size < 20 or np.abs(blk).max() > 0.5 * lim), name


def test_kl_jsons_carry_the_reference_sweep_axes():
    """jsonfiles/agent/{reverse,forward}_kl.json: 36 and 27 settings, first key varying fastest (utils/main_utils.py)."""
    import json
    import os
    from rlcontrol_amd.utils.main_utils import get_sweep_parameters
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "jsonfiles", "agent")
    from collections import OrderedDict
    for fn, agent, n in (("reverse_kl.json", "ReverseKL", 36), ("forward_kl.json", "ForwardKL", 27)):
        with open(os.path.join(root, fn)) as f:
            js = json.load(f, object_pairs_hook=OrderedDict)
        assert js["agent"] == agent
        p0, total = get_sweep_parameters(js["sweeps"], 0)
        assert total == n and p0["N_param"] == 64 and p0["optim_type"] == "intg" and p0["q_update_type"] == "non_sac"
    # the setting the reference's notebook ranks first for ForwardKL (plots.ipynb:90): index 18
    p18, _ = get_sweep_parameters(js["sweeps"], 18)
    assert (p18["pi_lr"], p18["qf_vf_lr"], p18["entropy_scale"]) == (1e-3, 1e-2, 0.001)
