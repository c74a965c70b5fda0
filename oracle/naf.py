"""ctypes front-end of oracle/naf_oracle.c (test infrastructure; see oracle/__init__.py).
Reference lines restated: agents/NAF.py:24-75; agents/network/naf_network.py:49-176."""
import ctypes
from collections import OrderedDict

import numpy as np

from .ddpg import lib, _fp, _dp


class NafDims(object):
    """(state_dim, action_dim, l1_dim, l2_dim)"""

    def __init__(self, S, A, L1, L2):
        self.t = (int(S), int(A), int(L1), int(L2))

    def layout(self):
        S, A, L1, L2 = self.t
        segs = [("W1", (S, L1)), ("b1", (L1,)), ("Wa2", (L1, L2)), ("ba2", (L2,)), ("Wa3", (L2, A)), ("ba3", (A,)),
                ("Wv2", (L1, L2)), ("bv2", (L2,)), ("Wv3", (L2, 1)), ("bv3", (1,))]
        for c in range(A):
            segs += [("Wd%d" % c, (L1, 1)), ("bd%d" % c, (1,))]
        for c in range(A - 1):
            segs += [("Wn%d" % c, (L1, A - 1 - c)), ("bn%d" % c, (A - 1 - c,))]
        out, p = OrderedDict(), 0
        for name, shp in segs:
            out[name] = (p, shp)
            p += int(np.prod(shp))
        return out, p

    @property
    def P(self):
        return self.layout()[1]


def init_params(dims, seed):
    """tf.contrib.layers.fully_connected defaults (naf_network.py:81-107): Glorot-uniform weights
    U(+-sqrt(6/(fan_in+fan_out))), zero biases; value output weights U(+-3e-3) (:96).  numpy RandomState(seed)
    stands in for TF's stream (distribution parity only)."""
    rng = np.random.RandomState(seed)
    lay, P = dims.layout()
    th = np.zeros(P, np.float32)
    for name, (off, shp) in lay.items():
        n = int(np.prod(shp))
        if name.startswith("b"):
            continue
        lim = 3e-3 if name == "Wv3" else np.sqrt(6.0 / (shp[0] + shp[1]))
        th[off:off + n] = rng.uniform(-lim, lim, n)
    return th


class NAFOracle(object):
    def __init__(self, dims, theta, lr, tau, state_min, state_max, action_max, clip_state=True):
        self.d = dims
        P = dims.P
        self.theta = np.asarray(theta, np.float32).copy()
        self.theta_t = self.theta.copy()
        self.m = np.zeros(P, np.float32)
        self.v = np.zeros(P, np.float32)
        self.pw = np.array([0.9, 0.999], np.float32)
        self.lr, self.tau = float(lr), float(tau)
        self.smin = np.ascontiguousarray(state_min, np.float32)
        self.smax = np.ascontiguousarray(state_max, np.float32)
        self.amax = np.ascontiguousarray(action_max, np.float32)
        self.clip = 1 if clip_state else 0

    def act(self, states):
        S, A = self.d.t[0], self.d.t[1]
        s = np.ascontiguousarray(states, np.float32).reshape(-1, S)
        mu = np.zeros((s.shape[0], A), np.float32)
        lc = np.zeros((s.shape[0], A * (A + 1) // 2), np.float32)
        lib().naf_oracle_act(*[ctypes.c_int(x) for x in self.d.t], _fp(self.theta), _fp(s), ctypes.c_int(s.shape[0]),
                             ctypes.c_int(self.clip), _fp(self.smin), _fp(self.smax), _fp(self.amax), _fp(mu), _fp(lc))
        return mu, lc

    def update(self, s, a, s2, r, gam, taps=False):
        S, A = self.d.t[0], self.d.t[1]
        B = len(r)
        f = lambda x, shp: np.ascontiguousarray(x, np.float32).reshape(shp)
        s, s2, a = f(s, (B, S)), f(s2, (B, S)), f(a, (B, A))
        r, gam = np.ascontiguousarray(r, np.float64).reshape(B), np.ascontiguousarray(gam, np.float64).reshape(B)
        t = None
        if taps:
            t = {"q": np.zeros(B, np.float32), "y": np.zeros(B, np.float32), "V": np.zeros(B, np.float32),
                 "grads": np.zeros(self.d.P, np.float32)}
        lib().naf_oracle_update(*[ctypes.c_int(x) for x in self.d.t], ctypes.c_int(B), _fp(self.theta), _fp(self.theta_t),
                                _fp(self.m), _fp(self.v), _fp(self.pw), _fp(s), _fp(a), _dp(r), _fp(s2), _dp(gam),
                                ctypes.c_float(self.lr), ctypes.c_float(self.tau), ctypes.c_int(self.clip), _fp(self.smin),
                                _fp(self.smax), _fp(self.amax), _fp(t["q"]) if t else None, _fp(t["y"]) if t else None,
                                _fp(t["V"]) if t else None, _fp(t["grads"]) if t else None)
        return t
