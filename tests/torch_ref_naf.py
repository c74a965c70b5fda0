"""Independent float64 torch.autograd restatement of the reference NAF update (test infrastructure): forward graph
of agents/network/naf_network.py:79-123 as written (lists of per-column heads, tf.slice products), SUM loss (:53),
TF-1.15 Adam, assign_add Polyak (:62-63), float64 TD glue (agents/NAF.py:70)."""
import numpy as np
import torch


class TorchNAF(object):
    def __init__(self, dims, theta, lr, tau, smin, smax, amax, clip_state=True):
        S, A, L1, L2 = dims
        self.dims = dims
        self.names = ["W1", "b1", "Wa2", "ba2", "Wa3", "ba3", "Wv2", "bv2", "Wv3", "bv3"]
        shapes = [(S, L1), (L1,), (L1, L2), (L2,), (L2, A), (A,), (L1, L2), (L2,), (L2, 1), (1,)]
        for c in range(A):
            self.names += ["Wd%d" % c, "bd%d" % c]
            shapes += [(L1, 1), (1,)]
        for c in range(A - 1):
            self.names += ["Wn%d" % c, "bn%d" % c]
            shapes += [(L1, A - 1 - c), (A - 1 - c,)]
        self.p, self.pt = {}, {}
        off = 0
        for n, shp in zip(self.names, shapes):
            k = int(np.prod(shp))
            w = torch.tensor(np.asarray(theta[off:off + k], np.float64).reshape(shp))
            self.p[n] = w.clone().requires_grad_(True)
            self.pt[n] = w.clone()
            off += k
        self.m = {n: torch.zeros_like(self.p[n]) for n in self.names}
        self.v = {n: torch.zeros_like(self.p[n]) for n in self.names}
        self.b1p, self.b2p = 0.9, 0.999
        self.lr, self.tau = lr, tau
        self.smin = torch.tensor(np.asarray(smin, np.float64))
        self.smax = torch.tensor(np.asarray(smax, np.float64))
        self.amax = torch.tensor(np.asarray(amax, np.float64))
        self.clip = clip_state

    def _x(self, s):
        x = torch.as_tensor(np.asarray(s, np.float32).astype(np.float64))
        return torch.max(torch.min(x, self.smax), self.smin) if self.clip else x

    def _net(self, P, x, action):
        A = self.dims[1]
        h1 = torch.relu(x @ P["W1"] + P["b1"])
        best = torch.tanh(torch.relu(h1 @ P["Wa2"] + P["ba2"]) @ P["Wa3"] + P["ba3"]) * self.amax
        value = torch.relu(h1 @ P["Wv2"] + P["bv2"]) @ P["Wv3"] + P["bv3"]
        if action is None:
            return best, value, None
        diff = action - best
        diag = [torch.exp(torch.clamp(h1 @ P["Wd%d" % c] + P["bd%d" % c], -5.0, 5.0)) for c in range(A)]
        nond = [h1 @ P["Wn%d" % c] + P["bn%d" % c] for c in range(A - 1)]
        cols = [torch.cat((diag[c], nond[c]), 1) for c in range(A - 1)] + [diag[-1]]
        prod = torch.cat([torch.sum(diff[:, c:] * cols[c], 1, keepdim=True) for c in range(A)], 1)
        adv = -0.5 * torch.sum(prod * prod, 1, keepdim=True)
        return best, value, value + adv

    def update(self, s, a, s2, r, gam):
        B = len(r)
        A = self.dims[1]
        x, x2 = self._x(s), self._x(s2)
        a = torch.as_tensor(np.asarray(a, np.float32).astype(np.float64)).reshape(B, A)
        with torch.no_grad():
            _, vt, _ = self._net(self.pt, x2, None)
        y = torch.as_tensor(np.asarray(r, np.float64) + np.asarray(gam, np.float64) * vt.numpy()[:, 0])
        _, V, q = self._net(self.p, x, a)
        loss = torch.sum((y - q[:, 0]) ** 2)
        gl = torch.autograd.grad(loss, [self.p[n] for n in self.names])
        taps = {"q": q.detach().numpy()[:, 0].copy(), "y": y.numpy().copy(), "V": V.detach().numpy()[:, 0].copy(),
                "grads": dict(zip(self.names, [g.numpy().copy() for g in gl]))}
        lr_t = self.lr * np.sqrt(1.0 - self.b2p) / (1.0 - self.b1p)
        with torch.no_grad():
            for n, g in zip(self.names, gl):
                self.m[n] += (g - self.m[n]) * 0.1
                self.v[n] += (g * g - self.v[n]) * 0.001
                self.p[n] -= (self.m[n] * lr_t) / (torch.sqrt(self.v[n]) + 1e-8)
            self.b1p *= 0.9
            self.b2p *= 0.999
            for n in self.names:
                self.pt[n] += self.tau * (self.p[n] - self.pt[n])
        return taps

    def blob(self, target=False):
        P = self.pt if target else self.p
        return np.concatenate([P[n].detach().numpy().reshape(-1) for n in self.names])
