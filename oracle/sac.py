"""ctypes front-end of oracle/sac_oracle.c (test infrastructure; see oracle/__init__.py).
Reference lines restated: agents/SoftActorCritic.py:55-126; agents/network/sac_network.py:47-136,152-307."""
import ctypes
from collections import OrderedDict

import numpy as np

from .ddpg import lib, _fp


class SacDims(object):
    """(state_dim, action_dim, actor_l1_dim, actor_l2_dim, critic_l1_dim, critic_l2_dim)"""

    def __init__(self, S, A, L1A, L2A, L1C, L2C):
        self.t = (int(S), int(A), int(L1A), int(L2A), int(L1C), int(L2C))

    def tuple(self):
        return self.t

    def layout(self):
        S, A, L1A, L2A, L1C, L2C = self.t
        out, p = OrderedDict(), 0
        for name, shp in (("pW1", (S, L1A)), ("pb1", (L1A,)), ("pW2", (L1A, L2A)), ("pb2", (L2A,)),
                          ("pWm", (L2A, A)), ("pbm", (A,)), ("pWs", (L2A, A)), ("pbs", (A,)),
                          ("qW1", (S, L1C)), ("qb1", (L1C,)), ("qW2", (L1C + A, L2C)), ("qb2", (L2C,)),
                          ("qW3", (L2C, 1)), ("qb3", (1,)),
                          ("vW1", (S, L1C)), ("vb1", (L1C,)), ("vW2", (L1C, L2C)), ("vb2", (L2C,)),
                          ("vW3", (L2C, 1)), ("vb3", (1,))):
            out[name] = (p, shp)
            p += int(np.prod(shp))
        return out, p

    @property
    def P(self):
        return self.layout()[1]


def init_params(dims, seed):
    """sac_network.py initialiser families: hidden W,b and the mu head ~ U(+-sqrt(3/fan_in)) (:178-270);
    log_std head W ~ U(0,1), b ~ U(+-3e-3) (:273-280); Q/V output layers ~ U(+-3e-3) (:198-201,227-230).
    numpy RandomState(seed) stands in for TF's unreproducible stream (distribution parity only)."""
    rng = np.random.RandomState(seed)
    lay, P = dims.layout()
    th = np.zeros(P, np.float32)
    for name, (off, shp) in lay.items():
        n = int(np.prod(shp))
        if name == "pWs":
            th[off:off + n] = rng.uniform(0.0, 1.0, n)
        elif name in ("pbs", "qW3", "qb3", "vW3", "vb3"):
            th[off:off + n] = rng.uniform(-3e-3, 3e-3, n)
        else:
            lim = np.sqrt(3.0 / shp[0])
            th[off:off + n] = rng.uniform(-lim, lim, n)
    return th


class SACOracle(object):
    def __init__(self, dims, theta, pi_lr, qv_lr, alpha, tau, smin0, smax0, amax0, clip_state=True):
        self.d = dims
        P = dims.P
        self.theta = np.asarray(theta, np.float32).copy()
        self.theta_t = self.theta.copy()                  # init_target_network (sac_network.py:75-76)
        self.m = np.zeros(P, np.float32)
        self.v = np.zeros(P, np.float32)
        self.pw = np.array([0.9, 0.999, 0.9, 0.999], np.float32)
        self.pi_lr, self.qv_lr, self.alpha, self.tau = float(pi_lr), float(qv_lr), float(alpha), float(tau)
        self.smin0, self.smax0, self.amax0, self.clip = float(smin0), float(smax0), float(amax0), 1 if clip_state else 0

    def act(self, states, eps=None):
        S, A = self.d.t[0], self.d.t[1]
        s = np.ascontiguousarray(states, np.float32).reshape(-1, S)
        out = np.zeros((s.shape[0], A), np.float32)
        e = None if eps is None else np.ascontiguousarray(eps, np.float32).reshape(s.shape[0], A)
        lib().sac_oracle_act(*[ctypes.c_int(x) for x in self.d.t], _fp(self.theta), _fp(s), ctypes.c_int(s.shape[0]),
                             ctypes.c_int(self.clip), ctypes.c_float(self.smin0), ctypes.c_float(self.smax0),
                             ctypes.c_float(self.amax0), _fp(e) if e is not None else None, _fp(out))
        return out

    def update(self, s, a, s2, r, gam, eps, taps=False):
        S, A = self.d.t[0], self.d.t[1]
        B = len(r)
        f = lambda x, shp: np.ascontiguousarray(x, np.float32).reshape(shp)
        s, s2, a, eps = f(s, (B, S)), f(s2, (B, S)), f(a, (B, A)), f(eps, (B, A))
        r, gam = f(r, (B,)), f(gam, (B,))
        t = None
        if taps:
            t = {"q": np.zeros(B, np.float32), "v": np.zeros(B, np.float32), "logp": np.zeros(B, np.float32),
                 "q_pi": np.zeros(B, np.float32), "loss": np.zeros(3, np.float32), "grads": np.zeros(self.d.P, np.float32)}
        cf = ctypes.c_float
        lib().sac_oracle_update(*[ctypes.c_int(x) for x in self.d.t], ctypes.c_int(B), _fp(self.theta), _fp(self.theta_t),
                                _fp(self.m), _fp(self.v), _fp(self.pw), _fp(s), _fp(a), _fp(r), _fp(s2), _fp(gam), _fp(eps),
                                cf(self.pi_lr), cf(self.qv_lr), cf(self.alpha), cf(self.tau), ctypes.c_int(self.clip),
                                cf(self.smin0), cf(self.smax0), cf(self.amax0),
                                _fp(t["q"]) if t else None, _fp(t["v"]) if t else None, _fp(t["logp"]) if t else None,
                                _fp(t["q_pi"]) if t else None, _fp(t["loss"]) if t else None, _fp(t["grads"]) if t else None)
        return t
