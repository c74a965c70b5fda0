"""CPU oracle of the NAF update with ``norm_type: layer`` -- TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

oracle/naf_oracle.c restates the shipped configuration (norm_type 'input_norm': activation only) in C with hand-written
back-propagation; the layer-norm variant (agents/network/base_network.py:53-56: tf.contrib.layers.layer_norm(center,
scale) before the relu of the trunk, the action branch and the value branch, naf_network.py:83,87,93) is restated here on
torch tensors with autograd forming the gradients -- fp32 by default (the oracle the HIP kernel is held to), float64 on
request (the twin the fp32 run is held to).  With norm_type 'input_norm' it must agree with oracle/naf_oracle.c
(tests/test_naf.py).  Everything else follows naf_oracle.c: y = float32(r + gamma V'(s')) formed in float64
(agents/NAF.py:70), loss = SUM (y - Q)^2 (naf_network.py:53-54), one TF-1.15 Adam with running beta powers, Polyak.
Parity status: "parity unpinned" (TensorFlow absent; no reference fixture exercises layer norm).
"""
from collections import OrderedDict

import numpy as np
import torch

LN_EPS = 1e-12      # tf.contrib.layers.layer_norm: variance_epsilon of tf.nn.batch_normalization


def layout(dims, norm):
    """name -> (offset, shape): variable creation order of naf_network.py:79-107, each normalised hidden layer followed
    by its layer-norm beta, gamma"""
    S, A, L1, L2 = dims
    ln = lambda tag, n: [("L%sb" % tag, (n,)), ("L%sg" % tag, (n,))] if norm else []
    spec = [("W1", (S, L1)), ("b1", (L1,))] + ln("1", L1) + [("Wa2", (L1, L2)), ("ba2", (L2,))] + ln("a2", L2) + \
           [("Wa3", (L2, A)), ("ba3", (A,)), ("Wv2", (L1, L2)), ("bv2", (L2,))] + ln("v2", L2) + \
           [("Wv3", (L2, 1)), ("bv3", (1,))]
    for c in range(A):
        spec += [("Wd%d" % c, (L1, 1)), ("bd%d" % c, (1,))]
    for c in range(A - 1):
        spec += [("Wn%d" % c, (L1, A - 1 - c)), ("bn%d" % c, (A - 1 - c,))]
    out, p = OrderedDict(), 0
    for name, shp in spec:
        out[name] = (p, shp)
        p += int(np.prod(shp))
    return out, p


def init_params(dims, seed, norm):
    """naf_network.py's initialiser families (see oracle/naf.py) + layer-norm beta 0, gamma 1"""
    rng = np.random.RandomState(seed)
    lay, P = layout(dims, norm)
    th = np.zeros(P, np.float32)
    for name, (off, shp) in lay.items():
        n = int(np.prod(shp))
        if name.startswith("L"):
            th[off:off + n] = 1.0 if name.endswith("g") else 0.0
        elif name.startswith("b"):
            continue
        else:
            lim = 3e-3 if name == "Wv3" else np.sqrt(6.0 / (shp[0] + shp[1]))
            th[off:off + n] = rng.uniform(-lim, lim, n)
    return th


class NafVariantOracle(object):
    def __init__(self, dims, theta, lr, tau, state_min, state_max, action_max, norm_type="layer", clip_state=True,
                 dtype=torch.float32):
        assert norm_type in ("input_norm", "layer")
        self.dims, self.norm, self.dt = tuple(int(x) for x in dims), norm_type == "layer", dtype
        self.lay, self.P = layout(self.dims, self.norm)
        self.theta = torch.tensor(np.asarray(theta, np.float32).copy()).to(dtype)
        self.theta_t = self.theta.clone()
        self.m = torch.zeros(self.P, dtype=dtype)
        self.v = torch.zeros(self.P, dtype=dtype)
        self.pw = np.array([0.9, 0.999], np.float32)
        self.lr, self.tau, self.clip = float(lr), float(tau), bool(clip_state)
        S, A = self.dims[0], self.dims[1]
        bc = lambda v, n: torch.tensor(np.broadcast_to(np.asarray(v, np.float32).reshape(-1), (n,)).copy()).to(dtype)
        self.smin, self.smax, self.amax = bc(state_min, S), bc(state_max, S), bc(action_max, A)

    def _views(self, flat):
        return {k: flat[o:o + int(np.prod(s))].view(*s) for k, (o, s) in self.lay.items()}

    def _act(self, P, tag, z):
        if self.norm:
            mean = z.mean(-1, keepdim=True)                   # tf.nn.moments over the features: biased variance
            var = ((z - mean) ** 2).mean(-1, keepdim=True)
            z = (z - mean) / torch.sqrt(var + LN_EPS) * P["L%sg" % tag] + P["L%sb" % tag]
        return torch.relu(z)

    def _clip(self, x):
        return torch.max(torch.min(x, self.smax), self.smin) if self.clip else x

    def _t(self, x, shape):
        return torch.as_tensor(np.asarray(x, np.float32)).to(self.dt).reshape(shape)

    def _heads(self, P, x):
        """mu [B,A] (scaled), V [B,1], the A columns of L (column c: [B, A-c])"""
        A = self.dims[1]
        h1 = self._act(P, "1", x @ P["W1"] + P["b1"])
        ha = self._act(P, "a2", h1 @ P["Wa2"] + P["ba2"])
        mu = torch.tanh(ha @ P["Wa3"] + P["ba3"]) * self.amax
        hv = self._act(P, "v2", h1 @ P["Wv2"] + P["bv2"])
        V = hv @ P["Wv3"] + P["bv3"]
        cols = []
        for c in range(A):
            diag = torch.exp(torch.clamp(h1 @ P["Wd%d" % c] + P["bd%d" % c], -5.0, 5.0))
            cols.append(torch.cat([diag, h1 @ P["Wn%d" % c] + P["bn%d" % c]], 1) if c < A - 1 else diag)
        return mu, V, cols

    def act(self, states):
        """greedy action [n,A] and the L columns [n, A(A+1)/2] (naf_network.py:144-158)"""
        S = self.dims[0]
        with torch.no_grad():
            mu, _, cols = self._heads(self._views(self.theta), self._clip(self._t(states, (-1, S))))
            return mu.to(torch.float32).numpy(), torch.cat(cols, 1).to(torch.float32).numpy()

    def update(self, s, a, s2, r, gam, taps=False):
        S, A = self.dims[0], self.dims[1]
        B = len(np.reshape(r, -1))
        s, s2, a = self._t(s, (B, S)), self._t(s2, (B, S)), self._t(a, (B, A))
        theta = self.theta.clone().requires_grad_(True)
        P, PT = self._views(theta), self._views(self.theta_t)
        with torch.no_grad():
            _, Vt, _ = self._heads(PT, self._clip(s2))
            y64 = np.asarray(r, np.float64).reshape(B) + np.asarray(gam, np.float64).reshape(B) * \
                Vt.reshape(B).to(torch.float64).numpy()
            y = torch.as_tensor(y64.astype(np.float32) if self.dt == torch.float32 else y64).to(self.dt).reshape(B, 1)
        mu, V, cols = self._heads(P, self._clip(s))
        diff = a - mu
        adv = 0.0
        for c in range(A):
            pc = torch.sum(diff[:, c:] * cols[c], 1, keepdim=True)
            adv = adv + pc * pc
        q = V + (-0.5 * adv)
        loss = torch.sum((y - q) ** 2)
        g = torch.autograd.grad(loss, theta)[0]
        with torch.no_grad():
            b1p, b2p = np.float32(self.pw[0]), np.float32(self.pw[1])
            lr_t = float(np.float32(self.lr) * np.sqrt(np.float32(1) - b2p) / (np.float32(1) - b1p))
            self.m += (g - self.m) * (1 - 0.9)
            self.v += (g * g - self.v) * (1 - 0.999)
            self.theta -= (self.m * lr_t) / (torch.sqrt(self.v) + 1e-8)
            self.pw *= np.array([0.9, 0.999], np.float32)
            self.theta_t = (1 - self.tau) * self.theta_t + self.tau * self.theta
        if not taps:
            return None
        out = {"q": q, "y": y, "V": V, "grads": g}
        return {k: t.detach().to(torch.float64).reshape(-1).numpy().copy() for k, t in out.items()}
