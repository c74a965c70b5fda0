"""Independent float64 torch.autograd restatement of the reference SAC-v1 update (test infrastructure).
Built from the forward graph / losses exactly as agents/network/sac_network.py:47-136,152-307 writes them --
including the [B] vs [B,1] broadcast of Q9 -- and TF-1.15 Adam; shares no code with oracle/sac_oracle.c."""
import numpy as np
import torch

EPS = 1e-6


class TorchSAC(object):
    NAMES = ["pW1", "pb1", "pW2", "pb2", "pWm", "pbm", "pWs", "pbs", "qW1", "qb1", "qW2", "qb2", "qW3", "qb3",
             "vW1", "vb1", "vW2", "vb2", "vW3", "vb3"]

    def __init__(self, dims, theta, pi_lr, qv_lr, alpha, tau, smin0, smax0, amax0, clip_state=True):
        S, A, L1A, L2A, L1C, L2C = dims
        self.dims = dims
        shapes = [(S, L1A), (L1A,), (L1A, L2A), (L2A,), (L2A, A), (A,), (L2A, A), (A,),
                  (S, L1C), (L1C,), (L1C + A, L2C), (L2C,), (L2C, 1), (1,),
                  (S, L1C), (L1C,), (L1C, L2C), (L2C,), (L2C, 1), (1,)]
        self.p, self.pt = {}, {}
        off = 0
        for n, shp in zip(self.NAMES, shapes):
            k = int(np.prod(shp))
            w = torch.tensor(np.asarray(theta[off:off + k], np.float64).reshape(shp))
            self.p[n] = w.clone().requires_grad_(True)
            self.pt[n] = w.clone()
            off += k
        self.pi_vars = self.NAMES[:8]
        self.val_vars = self.NAMES[8:]
        self.opt = {t: {"m": {n: torch.zeros_like(self.p[n]) for n in names}, "v": {n: torch.zeros_like(self.p[n]) for n in names},
                        "b1p": 0.9, "b2p": 0.999} for t, names in (("pi", self.pi_vars), ("val", self.val_vars))}
        self.lr = {"pi": pi_lr, "val": qv_lr}
        self.alpha, self.tau = alpha, tau
        self.smin0, self.smax0, self.amax0, self.clip = smin0, smax0, amax0, clip_state

    def _t(self, x, shape):
        return torch.as_tensor(np.asarray(x, np.float32).astype(np.float64)).reshape(shape)

    def _clip(self, x):
        return torch.clamp(x, self.smin0, self.smax0) if self.clip else x

    def _adam(self, tag, grads):
        o = self.opt[tag]
        lr_t = self.lr[tag] * np.sqrt(1.0 - o["b2p"]) / (1.0 - o["b1p"])
        with torch.no_grad():
            for n, g in grads.items():
                o["m"][n] += (g - o["m"][n]) * (1 - 0.9)
                o["v"][n] += (g * g - o["v"][n]) * (1 - 0.999)
                self.p[n] -= (o["m"][n] * lr_t) / (torch.sqrt(o["v"][n]) + 1e-8)
        o["b1p"] *= 0.9
        o["b2p"] *= 0.999

    def _qf(self, P, s, a):
        h = torch.relu(s @ P["qW1"] + P["qb1"])
        h = torch.relu(torch.cat([h, a], 1) @ P["qW2"] + P["qb2"])
        return h @ P["qW3"] + P["qb3"]

    def _vf(self, P, xc):
        h = torch.relu(xc @ P["vW1"] + P["vb1"])
        h = torch.relu(h @ P["vW2"] + P["vb2"])
        return h @ P["vW3"] + P["vb3"]

    def update(self, s, a, s2, r, gam, eps):
        S, A = self.dims[0], self.dims[1]
        B = len(r)
        P = self.p
        s, s2 = self._t(s, (B, S)), self._t(s2, (B, S))
        a, eps = self._t(a, (B, A)), self._t(eps, (B, A))
        r, gam = self._t(r, (B, 1)), self._t(gam, (B, 1))
        xc, x2c = self._clip(s), self._clip(s2)
        h = torch.relu(torch.relu(xc @ P["pW1"] + P["pb1"]) @ P["pW2"] + P["pb2"])
        mu = h @ P["pWm"] + P["pbm"]
        log_std = -20 + 0.5 * (2 - (-20)) * (torch.tanh(h @ P["pWs"] + P["pbs"]) + 1)
        std = torch.exp(log_std)
        u = mu + eps * std
        logp = torch.sum(-0.5 * (((u - mu) / (torch.exp(log_std) + EPS)) ** 2 + 2 * log_std + np.log(2 * np.pi)), 1)
        pit = torch.tanh(u)
        x = 1 - pit ** 2
        clipped = x + ((1 - x) * (x > 1).double() + (0 - x) * (x < 0).double()).detach()
        logp = logp - torch.sum(torch.log(clipped + 1e-6), 1)               # shape [B]
        pi = pit * self.amax0
        q = self._qf(P, s, a)                                                 # [B,1]
        q_pi = self._qf(P, s, pi)
        v = self._vf(P, xc)
        v_targ = self._vf(self.pt, x2c)
        q_backup = (r + gam * v_targ).detach()
        v_backup = (q_pi - self.alpha * logp).detach()                        # [B,B]  (Q9)
        pi_loss = torch.mean(self.alpha * logp - q_pi)                        # mean over [B,B]
        q_loss = 0.5 * torch.mean((q_backup - q) ** 2)
        v_loss = 0.5 * torch.mean((v_backup - v) ** 2)
        taps = {"q": q.detach().numpy()[:, 0].copy(), "v": v.detach().numpy()[:, 0].copy(),
                "logp": logp.detach().numpy().copy(), "q_pi": q_pi.detach().numpy()[:, 0].copy(),
                "loss": np.array([pi_loss.item(), q_loss.item(), v_loss.item()])}
        gpi = torch.autograd.grad(pi_loss, [P[n] for n in self.pi_vars], retain_graph=True)
        gval = torch.autograd.grad(q_loss + v_loss, [P[n] for n in self.val_vars])
        taps["grads"] = dict(zip(self.pi_vars + self.val_vars, [g.numpy().copy() for g in list(gpi) + list(gval)]))
        self._adam("pi", dict(zip(self.pi_vars, gpi)))
        self._adam("val", dict(zip(self.val_vars, gval)))
        with torch.no_grad():
            for n in self.NAMES:
                self.pt[n] = (1 - self.tau) * self.pt[n] + self.tau * self.p[n]
        return taps

    def blob(self, target=False):
        P = self.pt if target else self.p
        return np.concatenate([P[n].detach().numpy().reshape(-1) for n in self.NAMES])
