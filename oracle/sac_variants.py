"""CPU oracle of the SoftActorCritic update with ``norm_type: layer`` -- TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

oracle/sac_oracle.c restates the shipped configuration (norm_type 'input_norm': activation only) in C with hand-written
back-propagation; the layer-norm variant (agents/network/base_network.py:53-56: tf.contrib.layers.layer_norm(center,
scale) before every hidden relu of pi, qf and vf, sac_network.py:185,197,215,225,247,259) is restated here on torch
tensors with autograd forming the gradients -- fp32 by default (the oracle the HIP kernel is held to), float64 on request
(the twin the fp32 run is held to).  With norm_type 'input_norm' it must agree with oracle/sac_oracle.c (tests/test_sac.py).
Everything else follows sac_oracle.c: the Q9 broadcast, scalar state clip, TF-1.15 Adam with running beta powers.
Parity status: "parity unpinned" (TensorFlow absent; no reference fixture exercises layer norm).
"""
from collections import OrderedDict

import numpy as np
import torch

EPS = 1e-6
LN_EPS = 1e-12      # tf.contrib.layers.layer_norm: variance_epsilon of tf.nn.batch_normalization


def layout(dims, norm):
    """name -> (offset, shape): variable creation order, each hidden layer followed by its layer-norm beta, gamma"""
    S, A, L1A, L2A, L1C, L2C = dims
    spec = []
    for pre, w1, w2, l1, l2 in (("p", (S, L1A), (L1A, L2A), L1A, L2A), ("q", (S, L1C), (L1C + A, L2C), L1C, L2C),
                                ("v", (S, L1C), (L1C, L2C), L1C, L2C)):
        spec += [(pre + "W1", w1), (pre + "b1", (l1,))]
        if norm:
            spec += [(pre + "L1b", (l1,)), (pre + "L1g", (l1,))]
        spec += [(pre + "W2", w2), (pre + "b2", (l2,))]
        if norm:
            spec += [(pre + "L2b", (l2,)), (pre + "L2g", (l2,))]
        spec += [("pWm", (L2A, A)), ("pbm", (A,)), ("pWs", (L2A, A)), ("pbs", (A,))] if pre == "p" else \
                [(pre + "W3", (l2, 1)), (pre + "b3", (1,))]
    out, p = OrderedDict(), 0
    for name, shp in spec:
        out[name] = (p, shp)
        p += int(np.prod(shp))
    return out, p


def init_params(dims, seed, norm):
    """sac_network.py's initialiser families (see oracle/sac.py) + layer-norm beta 0, gamma 1"""
    rng = np.random.RandomState(seed)
    lay, P = layout(dims, norm)
    th = np.zeros(P, np.float32)
    for name, (off, shp) in lay.items():
        n = int(np.prod(shp))
        if name[1] == "L":
            th[off:off + n] = 1.0 if name.endswith("g") else 0.0
        elif name == "pWs":
            th[off:off + n] = rng.uniform(0.0, 1.0, n)
        elif name in ("pbs", "qW3", "qb3", "vW3", "vb3"):
            th[off:off + n] = rng.uniform(-3e-3, 3e-3, n)
        else:
            lim = np.sqrt(3.0 / shp[0])
            th[off:off + n] = rng.uniform(-lim, lim, n)
    return th


class SacVariantOracle(object):
    def __init__(self, dims, theta, pi_lr, qv_lr, alpha, tau, smin0, smax0, amax0, norm_type="layer", clip_state=True,
                 dtype=torch.float32):
        assert norm_type in ("input_norm", "layer")
        self.dims, self.norm, self.dt = tuple(int(x) for x in dims), norm_type == "layer", dtype
        self.lay, self.P = layout(self.dims, self.norm)
        self.theta = torch.tensor(np.asarray(theta, np.float32).copy()).to(dtype)
        self.theta_t = self.theta.clone()
        self.m = torch.zeros(self.P, dtype=dtype)
        self.v = torch.zeros(self.P, dtype=dtype)
        self.pw = np.array([0.9, 0.999, 0.9, 0.999], np.float32)
        self.pi_lr, self.qv_lr, self.alpha, self.tau = float(pi_lr), float(qv_lr), float(alpha), float(tau)
        self.smin0, self.smax0, self.amax0, self.clip = float(smin0), float(smax0), float(amax0), bool(clip_state)
        self.pi_end = self.lay["qW1"][0]

    def _views(self, flat):
        return {k: flat[o:o + int(np.prod(s))].view(*s) for k, (o, s) in self.lay.items()}

    def _act(self, P, pre, layer, z):
        if self.norm:
            mean = z.mean(-1, keepdim=True)                   # tf.nn.moments over the features: biased variance
            var = ((z - mean) ** 2).mean(-1, keepdim=True)
            z = (z - mean) / torch.sqrt(var + LN_EPS) * P[pre + "L%dg" % layer] + P[pre + "L%db" % layer]
        return torch.relu(z)

    def _pi_hidden(self, P, xc):
        h = self._act(P, "p", 1, xc @ P["pW1"] + P["pb1"])
        return self._act(P, "p", 2, h @ P["pW2"] + P["pb2"])

    def _qf(self, P, s, a):
        h = self._act(P, "q", 1, s @ P["qW1"] + P["qb1"])
        h = self._act(P, "q", 2, torch.cat([h, a], 1) @ P["qW2"] + P["qb2"])
        return h @ P["qW3"] + P["qb3"]

    def _vf(self, P, xc):
        h = self._act(P, "v", 1, xc @ P["vW1"] + P["vb1"])
        h = self._act(P, "v", 2, h @ P["vW2"] + P["vb2"])
        return h @ P["vW3"] + P["vb3"]

    def _clip(self, x):
        return torch.clamp(x, self.smin0, self.smax0) if self.clip else x

    def _t(self, x, shape):
        return torch.as_tensor(np.asarray(x, np.float32)).to(self.dt).reshape(shape)

    def act(self, states, eps=None):
        S, A = self.dims[0], self.dims[1]
        with torch.no_grad():
            P = self._views(self.theta)
            h = self._pi_hidden(P, self._clip(self._t(states, (-1, S))))
            u = h @ P["pWm"] + P["pbm"]
            if eps is not None:
                log_std = -20 + 0.5 * (2 - (-20)) * (torch.tanh(h @ P["pWs"] + P["pbs"]) + 1)
                u = u + self._t(eps, u.shape) * torch.exp(log_std)
            return (torch.tanh(u) * self.amax0).to(torch.float32).numpy()

    def update(self, s, a, s2, r, gam, eps, taps=False):
        S, A = self.dims[0], self.dims[1]
        B = len(np.reshape(r, -1))
        s, s2 = self._t(s, (B, S)), self._t(s2, (B, S))
        a, eps = self._t(a, (B, A)), self._t(eps, (B, A))
        r, gam = self._t(r, (B, 1)), self._t(gam, (B, 1))
        theta = self.theta.clone().requires_grad_(True)
        P, PT = self._views(theta), self._views(self.theta_t)
        xc, x2c = self._clip(s), self._clip(s2)
        h = self._pi_hidden(P, xc)
        mu = h @ P["pWm"] + P["pbm"]
        log_std = -20 + 0.5 * (2 - (-20)) * (torch.tanh(h @ P["pWs"] + P["pbs"]) + 1)
        std = torch.exp(log_std)
        u = mu + eps * std
        logp = torch.sum(-0.5 * (((u - mu) / (std + EPS)) ** 2 + 2 * log_std + np.log(2 * np.pi)), 1)
        pit = torch.tanh(u)
        x = 1 - pit ** 2
        clipped = x + ((1 - x) * (x > 1).to(self.dt) + (0 - x) * (x < 0).to(self.dt)).detach()
        logp = logp - torch.sum(torch.log(clipped + 1e-6), 1)               # [B]
        pi = pit * self.amax0
        q = self._qf(P, s, a)
        q_pi = self._qf(P, s, pi)
        v = self._vf(P, xc)
        v_targ = self._vf(PT, x2c)
        q_backup = (r + gam * v_targ).detach()
        v_backup = (q_pi - self.alpha * logp).detach()                        # [B,B]: quirk Q9
        pi_loss = torch.mean(self.alpha * logp - q_pi)
        q_loss = 0.5 * torch.mean((q_backup - q) ** 2)
        v_loss = 0.5 * torch.mean((v_backup - v) ** 2)
        g_pi = torch.autograd.grad(pi_loss, theta, retain_graph=True)[0]
        g_val = torch.autograd.grad(q_loss + v_loss, theta)[0]
        g = torch.cat([g_pi[:self.pi_end], g_val[self.pi_end:]])
        with torch.no_grad():
            for lo, hi, lr, k in ((0, self.pi_end, self.pi_lr, 0), (self.pi_end, self.P, self.qv_lr, 2)):
                b1p, b2p = np.float32(self.pw[k]), np.float32(self.pw[k + 1])
                lr_t = float(np.float32(lr) * np.sqrt(np.float32(1) - b2p) / (np.float32(1) - b1p))
                self.m[lo:hi] += (g[lo:hi] - self.m[lo:hi]) * (1 - 0.9)
                self.v[lo:hi] += (g[lo:hi] * g[lo:hi] - self.v[lo:hi]) * (1 - 0.999)
                self.theta[lo:hi] -= (self.m[lo:hi] * lr_t) / (torch.sqrt(self.v[lo:hi]) + 1e-8)
            self.pw *= np.array([0.9, 0.999, 0.9, 0.999], np.float32)
            self.theta_t = (1 - self.tau) * self.theta_t + self.tau * self.theta
        if not taps:
            return None
        out = {"q": q, "v": v, "logp": logp, "q_pi": q_pi, "grads": g, "loss": torch.stack([pi_loss, q_loss, v_loss])}
        return {k: t.detach().to(torch.float64).reshape(-1).numpy().copy() for k, t in out.items()}
