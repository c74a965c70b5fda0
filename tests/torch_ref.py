"""Second, independent restatement of the reference DDPG update: torch.autograd in float64.

Test infrastructure.  It shares no code with oracle/ddpg_oracle.c (which back-propagates by hand in
fp32): gradients here come from autograd on the forward graph exactly as the reference builds it
(agents/network/hydra_ddpg_network.py:36-37,71-75,97-142), the step order follows
agents/DDPG.py:74-95, Adam follows TF-1.15's ApplyAdam (epsilon outside the bias correction, Q2).
Agreement of the two to ~1e-6 is what stands in for the un-runnable TensorFlow path.
"""
import numpy as np
import torch


class TorchDDPG(object):
    def __init__(self, dims, theta, actor_lr, critic_lr, tau, state_min, state_max, action_max, clip_state=True,
                 dtype=torch.float64):
        self.S, self.A, self.H1, self.HA, self.HC = dims
        self.dtype = dtype
        self.names = ["W1", "b1", "Wa2", "ba2", "Wa3", "ba3", "Wc2", "bc2", "Wc3", "bc3"]
        shapes = [(self.S, self.H1), (self.H1,), (self.H1, self.HA), (self.HA,), (self.HA, self.A), (self.A,),
                  (self.H1 + self.A, self.HC), (self.HC,), (self.HC, 1), (1,)]
        self.p, self.pt = {}, {}
        off = 0
        for n, shp in zip(self.names, shapes):
            k = int(np.prod(shp))
            w = torch.tensor(np.asarray(theta[off:off + k], np.float64).reshape(shp), dtype=dtype)
            self.p[n] = w.clone().requires_grad_(True)
            self.pt[n] = w.clone()
            off += k
        self.actor_vars = ["W1", "b1", "Wa2", "ba2", "Wa3", "ba3"]
        self.critic_vars = ["W1", "b1", "Wc2", "bc2", "Wc3", "bc3"]
        self.opt = {}
        for tag, names in (("a", self.actor_vars), ("c", self.critic_vars)):
            self.opt[tag] = {"m": {n: torch.zeros_like(self.p[n]) for n in names},
                             "v": {n: torch.zeros_like(self.p[n]) for n in names},
                             "b1p": 0.9, "b2p": 0.999}
        self.lr = {"a": actor_lr, "c": critic_lr}
        self.tau = tau
        self.smin = torch.tensor(np.asarray(state_min, np.float64), dtype=dtype)
        self.smax = torch.tensor(np.asarray(state_max, np.float64), dtype=dtype)
        self.amax = torch.tensor(np.asarray(action_max, np.float64), dtype=dtype)
        self.clip = clip_state

    # ---- graph (hydra_ddpg_network.py:78-142) ----
    def _x(self, s):
        x = torch.as_tensor(np.asarray(s, np.float32).astype(np.float64), dtype=self.dtype)
        if self.clip:
            x = torch.max(torch.min(x, self.smax), self.smin)
        return x

    def _net(self, P, x, action):
        h1 = torch.relu(x @ P["W1"] + P["b1"])
        h2 = torch.relu(h1 @ P["Wa2"] + P["ba2"])
        mu = torch.tanh(h2 @ P["Wa3"] + P["ba3"])
        q = None
        if action is not None:
            g2 = torch.relu(torch.cat([h1, action], 1) @ P["Wc2"] + P["bc2"])
            q = g2 @ P["Wc3"] + P["bc3"]
        return mu, q

    def _adam(self, tag, grads):
        o = self.opt[tag]
        lr_t = self.lr[tag] * np.sqrt(1.0 - o["b2p"]) / (1.0 - o["b1p"])
        with torch.no_grad():
            for n, g in grads.items():
                if g is None:
                    continue
                o["m"][n] += (g - o["m"][n]) * (1 - 0.9)
                o["v"][n] += (g * g - o["v"][n]) * (1 - 0.999)
                self.p[n] -= (o["m"][n] * lr_t) / (torch.sqrt(o["v"][n]) + 1e-8)
        o["b1p"] *= 0.9
        o["b2p"] *= 0.999

    def act(self, s):
        with torch.no_grad():
            mu, _ = self._net(self.p, self._x(s), None)
            return (mu * self.amax).numpy()

    def qval(self, s, a):
        with torch.no_grad():
            a = torch.as_tensor(np.asarray(a, np.float32).astype(np.float64), dtype=self.dtype).reshape(-1, self.A)
            return self._net(self.p, self._x(s), a)[1].numpy()[:, 0]

    # ---- agents/DDPG.py:74-95 ----
    def update(self, s, a, s2, r, gam):
        B = len(r)
        x, x2 = self._x(s), self._x(s2)
        a = torch.as_tensor(np.asarray(a, np.float32).astype(np.float64), dtype=self.dtype).reshape(B, self.A)
        taps = {}
        with torch.no_grad():
            mu_t, _ = self._net(self.pt, x2, None)
            _, q_t = self._net(self.pt, x2, mu_t * self.amax)
        y = torch.as_tensor(np.asarray(r, np.float64).reshape(B, 1) + np.asarray(gam, np.float64).reshape(B, 1)
                            * q_t.numpy(), dtype=self.dtype)
        taps["y"] = y.numpy()[:, 0].copy()
        # critic step
        _, q = self._net(self.p, x, a)
        taps["q"] = q.detach().numpy()[:, 0].copy()
        loss = torch.mean((y - q) ** 2)
        gl = torch.autograd.grad(loss, [self.p[n] for n in self.critic_vars])
        taps["grads_c"] = dict(zip(self.critic_vars, [g.numpy().copy() for g in gl]))
        self._adam("c", dict(zip(self.critic_vars, gl)))
        # actor forward, dQ/da, actor step
        with torch.no_grad():
            mu, _ = self._net(self.p, x, None)
            a_out = (mu * self.amax)
        taps["a_out"] = a_out.numpy().copy()
        a_in = a_out.clone().requires_grad_(True)
        _, q2 = self._net(self.p, x, a_in)
        dqda = torch.autograd.grad(q2.sum(), a_in)[0]
        taps["dqda"] = dqda.numpy().copy()
        mu2, _ = self._net(self.p, x, None)
        ga = torch.autograd.grad(mu2, [self.p[n] for n in self.actor_vars], grad_outputs=-dqda)
        taps["grads_a"] = dict(zip(self.actor_vars, [g.numpy().copy() for g in ga]))
        self._adam("a", dict(zip(self.actor_vars, ga)))
        with torch.no_grad():
            for n in self.names:
                self.pt[n] += self.tau * (self.p[n] - self.pt[n])
        return taps

    def blob(self, target=False):
        P = self.pt if target else self.p
        return np.concatenate([P[n].detach().numpy().reshape(-1) for n in self.names])
