"""ctypes front-end of oracle/ddpg_variants_oracle.c (test infrastructure; see oracle/__init__.py): DDPG with
norm_type 'layer' (agents/network/base_network.py:53-56) and/or separate actor / critic networks
(agents/network/actor_network.py:73-96, critic_network.py:77-99)."""
import ctypes
from collections import OrderedDict

import numpy as np

from .ddpg import lib, _fp, _dp


class VDims(object):
    """(state_dim, action_dim, l1, actor_l2, critic_l2, layer_norm, separate_networks)"""

    def __init__(self, S, A, H1, HA, HC, norm=False, separate=False):
        self.S, self.A, self.H1, self.HA, self.HC = int(S), int(A), int(H1), int(HA), int(HC)
        self.norm, self.sep = bool(norm), bool(separate)

    def tuple(self):
        return (self.S, self.A, self.H1, self.HA, self.HC, int(self.norm), int(self.sep))

    def layout(self):
        """name -> (offset, shape), variable creation order (beta before gamma inside each LayerNorm scope)"""
        S, A, H1, HA, HC = self.S, self.A, self.H1, self.HA, self.HC
        ln = lambda tag, n: [(tag + "b", (n,)), (tag + "g", (n,))] if self.norm else []
        items = [("W1", (S, H1)), ("b1", (H1,))] + ln("l1", H1) + [("Wa2", (H1, HA)), ("ba2", (HA,))] + ln("l2", HA) + \
                [("Wa3", (HA, A)), ("ba3", (A,))]
        if self.sep:
            items += [("Wc1", (S, H1)), ("bc1", (H1,))] + ln("lc", H1)
        items += [("Wc2", (H1 + A, HC)), ("bc2", (HC,))] + ln("l3", HC) + [("Wc3", (HC, 1)), ("bc3", (1,))]
        out, p = OrderedDict(), 0
        for name, shp in items:
            out[name] = (p, shp)
            p += int(np.prod(shp))
        return out, p

    @property
    def P(self):
        return self.layout()[1]


def init_params(dims, seed):
    """as oracle.ddpg.init_params; LayerNorm beta = 0, gamma = 1 (tf.contrib.layers.layer_norm defaults)"""
    rng = np.random.RandomState(seed)
    lay, P = dims.layout()
    th = np.zeros(P, np.float32)
    for name, (off, shp) in lay.items():
        n = int(np.prod(shp))
        if name[0] == "l":
            th[off:off + n] = 1.0 if name.endswith("g") else 0.0
            continue
        lim = 3e-3 if name in ("Wa3", "ba3", "Wc3", "bc3") else np.sqrt(3.0 / shp[0])
        th[off:off + n] = rng.uniform(-lim, lim, n).astype(np.float32)
    return th


class DDPGVariantOracle(object):
    def __init__(self, dims, theta, actor_lr, critic_lr, tau, state_min, state_max, action_max, clip_state=True):
        self.d = dims
        P = dims.P
        assert theta.shape == (P,)
        self.theta = theta.astype(np.float32).copy()
        self.theta_t = self.theta.copy()
        self.m_a, self.v_a, self.m_c, self.v_c = (np.zeros(P, np.float32) for _ in range(4))
        self.pw = np.array([0.9, 0.999, 0.9, 0.999], np.float32)
        self.actor_lr, self.critic_lr, self.tau = float(actor_lr), float(critic_lr), float(tau)
        self.smin = np.ascontiguousarray(state_min, np.float32)
        self.smax = np.ascontiguousarray(state_max, np.float32)
        self.amax = np.ascontiguousarray(action_max, np.float32)
        self.clip = 1 if clip_state else 0

    def act(self, states):
        s = np.ascontiguousarray(states, np.float32).reshape(-1, self.d.S)
        out = np.zeros((s.shape[0], self.d.A), np.float32)
        lib().ddpg_variant_act(*[ctypes.c_int(v) for v in self.d.tuple()], _fp(self.theta), _fp(s), ctypes.c_int(s.shape[0]),
                               ctypes.c_int(self.clip), _fp(self.smin), _fp(self.smax), _fp(self.amax), _fp(out))
        return out

    def update(self, s, a, s2, r, gam, taps=False):
        B = len(r)
        s = np.ascontiguousarray(s, np.float32).reshape(B, self.d.S)
        a = np.ascontiguousarray(a, np.float32).reshape(B, self.d.A)
        s2 = np.ascontiguousarray(s2, np.float32).reshape(B, self.d.S)
        r = np.ascontiguousarray(r, np.float64).reshape(B)
        gam = np.ascontiguousarray(gam, np.float64).reshape(B)
        t = None
        if taps:
            P = self.d.P
            t = {"q": np.zeros(B, np.float32), "y": np.zeros(B, np.float32),
                 "a_out": np.zeros((B, self.d.A), np.float32), "dqda": np.zeros((B, self.d.A), np.float32),
                 "grads_c": np.zeros(P, np.float32), "grads_a": np.zeros(P, np.float32)}
        lib().ddpg_variant_update(
            *[ctypes.c_int(v) for v in self.d.tuple()], ctypes.c_int(B),
            _fp(self.theta), _fp(self.theta_t), _fp(self.m_a), _fp(self.v_a), _fp(self.m_c), _fp(self.v_c),
            _fp(self.pw), _fp(s), _fp(a), _dp(r), _fp(s2), _dp(gam),
            ctypes.c_float(self.actor_lr), ctypes.c_float(self.critic_lr), ctypes.c_float(self.tau),
            ctypes.c_int(self.clip), _fp(self.smin), _fp(self.smax), _fp(self.amax),
            _fp(t["q"]) if t else None, _fp(t["y"]) if t else None, _fp(t["a_out"]) if t else None,
            _fp(t["dqda"]) if t else None, _fp(t["grads_c"]) if t else None, _fp(t["grads_a"]) if t else None)
        return t
