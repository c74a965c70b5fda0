"""float64 autograd twin of oracle/ddpg_variants_oracle.c (test infrastructure): DDPG with tf.contrib layer_norm after
every hidden layer and/or separate actor / critic networks.  Gradients come from autograd on the forward graph as the
reference builds it (base_network.py:53-56, hydra_ddpg_network.py:97-142, actor_network.py:73-96,
critic_network.py:77-99); step order agents/DDPG.py:74-95; Adam = TF-1.15 ApplyAdam."""
import numpy as np
import torch


class TorchDDPGVariant(object):
    def __init__(self, vdims, theta, actor_lr, critic_lr, tau, state_min, state_max, action_max, clip_state=True):
        self.d = vdims
        lay, P = vdims.layout()
        self.names = list(lay)
        dt = torch.float64
        self.p, self.pt = {}, {}
        for n, (off, shp) in lay.items():
            w = torch.tensor(np.asarray(theta[off:off + int(np.prod(shp))], np.float64).reshape(shp), dtype=dt)
            self.p[n] = w.clone().requires_grad_(True)
            self.pt[n] = w.clone()
        first_critic = "Wc1" if vdims.sep else "Wc2"
        k = self.names.index(first_critic)
        self.actor_vars = self.names[:k]
        shared = [] if vdims.sep else [n for n in self.names[:self.names.index("Wa2")]]
        self.critic_vars = shared + self.names[k:]
        self.opt = {t: {"m": {n: torch.zeros_like(self.p[n]) for n in v}, "v": {n: torch.zeros_like(self.p[n]) for n in v},
                        "b1p": 0.9, "b2p": 0.999} for t, v in (("a", self.actor_vars), ("c", self.critic_vars))}
        self.lr = {"a": actor_lr, "c": critic_lr}
        self.tau = tau
        self.smin = torch.tensor(np.asarray(state_min, np.float64))
        self.smax = torch.tensor(np.asarray(state_max, np.float64))
        self.amax = torch.tensor(np.asarray(action_max, np.float64))
        self.clip = clip_state

    def _x(self, s):
        x = torch.as_tensor(np.asarray(s, np.float32).astype(np.float64))
        return torch.max(torch.min(x, self.smax), self.smin) if self.clip else x

    def _hidden(self, P, z, tag):
        if self.d.norm:                      # tf.contrib.layers.layer_norm: moments over the features, eps 1e-12
            mean = z.mean(1, keepdim=True)
            var = ((z - mean) ** 2).mean(1, keepdim=True)
            z = (z - mean) * torch.rsqrt(var + 1e-12) * P[tag + "g"] + P[tag + "b"]
        return torch.relu(z)

    def _actor(self, P, x):
        h1 = self._hidden(P, x @ P["W1"] + P["b1"], "l1")
        h2 = self._hidden(P, h1 @ P["Wa2"] + P["ba2"], "l2")
        return torch.tanh(h2 @ P["Wa3"] + P["ba3"])

    def _critic(self, P, x, action):
        if self.d.sep:
            c1 = self._hidden(P, x @ P["Wc1"] + P["bc1"], "lc")
        else:
            c1 = self._hidden(P, x @ P["W1"] + P["b1"], "l1")
        g2 = self._hidden(P, torch.cat([c1, action], 1) @ P["Wc2"] + P["bc2"], "l3")
        return g2 @ P["Wc3"] + P["bc3"]

    def _adam(self, tag, names, grads):
        o = self.opt[tag]
        lr_t = self.lr[tag] * np.sqrt(1.0 - o["b2p"]) / (1.0 - o["b1p"])
        with torch.no_grad():
            for n, g in zip(names, grads):
                if g is None:
                    continue
                o["m"][n] += (g - o["m"][n]) * (1 - 0.9)
                o["v"][n] += (g * g - o["v"][n]) * (1 - 0.999)
                self.p[n] -= (o["m"][n] * lr_t) / (torch.sqrt(o["v"][n]) + 1e-8)
        o["b1p"] *= 0.9
        o["b2p"] *= 0.999

    def update(self, s, a, s2, r, gam):
        B = len(r)
        x, x2 = self._x(s), self._x(s2)
        a = torch.as_tensor(np.asarray(a, np.float32).astype(np.float64)).reshape(B, self.d.A)
        taps = {}
        with torch.no_grad():
            q_t = self._critic(self.pt, x2, self._actor(self.pt, x2) * self.amax)
        y = torch.as_tensor(np.asarray(r, np.float64).reshape(B, 1) + np.asarray(gam, np.float64).reshape(B, 1) * q_t.numpy())
        taps["y"] = y.numpy()[:, 0].copy()
        q = self._critic(self.p, x, a)
        taps["q"] = q.detach().numpy()[:, 0].copy()
        gl = torch.autograd.grad(torch.mean((y - q) ** 2), [self.p[n] for n in self.critic_vars])
        taps["grads_c"] = dict(zip(self.critic_vars, [g.numpy().copy() for g in gl]))
        self._adam("c", self.critic_vars, gl)
        with torch.no_grad():
            a_out = self._actor(self.p, x) * self.amax
        taps["a_out"] = a_out.numpy().copy()
        a_in = a_out.clone().requires_grad_(True)
        dqda = torch.autograd.grad(self._critic(self.p, x, a_in).sum(), a_in)[0]
        taps["dqda"] = dqda.numpy().copy()
        ga = torch.autograd.grad(self._actor(self.p, x), [self.p[n] for n in self.actor_vars], grad_outputs=-dqda)
        taps["grads_a"] = dict(zip(self.actor_vars, [g.numpy().copy() for g in ga]))
        self._adam("a", self.actor_vars, ga)
        with torch.no_grad():
            for n in self.names:
                self.pt[n] += self.tau * (self.p[n] - self.pt[n])
        return taps

    def blob(self, target=False):
        P = self.pt if target else self.p
        return np.concatenate([P[n].detach().numpy().reshape(-1) for n in self.names])
