"""ctypes front-end of oracle/ddpg_oracle.c (test infrastructure; see oracle/__init__.py).

Reference lines restated: agents/DDPG.py:34-36,74-95; agents/network/hydra_ddpg_network.py:29-142.
"""
import ctypes
import os
import subprocess
from collections import OrderedDict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "ddpg_oracle.c")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    assert os.path.exists(src)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.ddpg_oracle_param_count.restype = ctypes.c_int
    return _LIB


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if a is not None else None


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)) if a is not None else None


class Dims(object):
    """(state_dim, action_dim, shared_l1_dim, actor_l2_dim, critic_l2_dim)"""

    def __init__(self, S, A, H1, HA, HC):
        self.S, self.A, self.H1, self.HA, self.HC = int(S), int(A), int(H1), int(HA), int(HC)

    def tuple(self):
        return (self.S, self.A, self.H1, self.HA, self.HC)

    def layout(self):
        """name -> (offset, shape) in variable-creation order (hydra_ddpg_network.py:100-140)"""
        S, A, H1, HA, HC = self.tuple()
        out = OrderedDict()
        p = 0
        for name, shp in (("W1", (S, H1)), ("b1", (H1,)), ("Wa2", (H1, HA)), ("ba2", (HA,)),
                          ("Wa3", (HA, A)), ("ba3", (A,)), ("Wc2", (H1 + A, HC)), ("bc2", (HC,)),
                          ("Wc3", (HC, 1)), ("bc3", (1,))):
            out[name] = (p, shp)
            p += int(np.prod(shp))
        return out, p

    @property
    def P(self):
        return self.layout()[1]


def init_params(dims, seed):
    """hydra_ddpg_network.py:101-105,112-116,122-124,129-133,138-140 initialiser FAMILIES:
    hidden W and b ~ U(+-sqrt(3/fan_in)) (variance_scaling_initializer(factor=1, FAN_IN, uniform);
    a 1-D bias of shape [n] has fan_in = n), output layers ~ U(+-3e-3).
    TF's own Philox stream cannot be reproduced without TF, so the draw uses numpy RandomState(seed):
    distribution parity only."""
    rng = np.random.RandomState(seed)
    lay, P = dims.layout()
    th = np.zeros(P, np.float32)
    for name, (off, shp) in lay.items():
        n = int(np.prod(shp))
        if name in ("Wa3", "ba3", "Wc3", "bc3"):
            lim = 3e-3
        else:
            fan_in = shp[0] if len(shp) > 1 else shp[0]
            lim = np.sqrt(3.0 / fan_in)
        th[off:off + n] = rng.uniform(-lim, lim, n).astype(np.float32)
    return th


class DDPGOracle(object):
    """State of one reference DDPG agent's networks + the two Adam optimizers."""

    def __init__(self, dims, theta, actor_lr, critic_lr, tau, state_min, state_max, action_max, clip_state=True):
        self.d = dims
        P = dims.P
        assert theta.shape == (P,)
        self.theta = theta.astype(np.float32).copy()
        self.theta_t = self.theta.copy()                    # init_target_network (hydra_ddpg_network.py:32)
        self.m_a = np.zeros(P, np.float32)
        self.v_a = np.zeros(P, np.float32)
        self.m_c = np.zeros(P, np.float32)
        self.v_c = np.zeros(P, np.float32)
        self.pw = np.array([0.9, 0.999, 0.9, 0.999], np.float32)   # beta powers start at beta (Q2)
        self.actor_lr, self.critic_lr, self.tau = float(actor_lr), float(critic_lr), float(tau)
        self.smin = np.ascontiguousarray(state_min, np.float32)
        self.smax = np.ascontiguousarray(state_max, np.float32)
        self.amax = np.ascontiguousarray(action_max, np.float32)
        self.clip = 1 if clip_state else 0

    def act(self, states, target=False):
        s = np.ascontiguousarray(states, np.float32).reshape(-1, self.d.S)
        out = np.zeros((s.shape[0], self.d.A), np.float32)
        th = self.theta_t if target else self.theta
        lib().ddpg_oracle_act(*[ctypes.c_int(v) for v in self.d.tuple()], _fp(th), _fp(s),
                              ctypes.c_int(s.shape[0]), ctypes.c_int(self.clip), _fp(self.smin),
                              _fp(self.smax), _fp(self.amax), _fp(out))
        return out

    def qval(self, states, actions, target=False):
        s = np.ascontiguousarray(states, np.float32).reshape(-1, self.d.S)
        a = np.ascontiguousarray(actions, np.float32).reshape(-1, self.d.A)
        out = np.zeros((s.shape[0],), np.float32)
        th = self.theta_t if target else self.theta
        lib().ddpg_oracle_qval(*[ctypes.c_int(v) for v in self.d.tuple()], _fp(th), _fp(s), _fp(a),
                               ctypes.c_int(s.shape[0]), ctypes.c_int(self.clip), _fp(self.smin),
                               _fp(self.smax), _fp(out))
        return out

    def update(self, s, a, s2, r, gam, taps=False):
        """update_network(state, action, next_state, reward, gamma) -- argument order of
        base_network_manager.py:78.  s,a,s2 are cast to fp32 (the placeholder feed); r, gamma stay float64."""
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
        lib().ddpg_oracle_update(
            *[ctypes.c_int(v) for v in self.d.tuple()], ctypes.c_int(B),
            _fp(self.theta), _fp(self.theta_t), _fp(self.m_a), _fp(self.v_a), _fp(self.m_c), _fp(self.v_c),
            _fp(self.pw), _fp(s), _fp(a), _dp(r), _fp(s2), _dp(gam),
            ctypes.c_float(self.actor_lr), ctypes.c_float(self.critic_lr), ctypes.c_float(self.tau),
            ctypes.c_int(self.clip), _fp(self.smin), _fp(self.smax), _fp(self.amax),
            _fp(t["q"]) if t else None, _fp(t["y"]) if t else None, _fp(t["a_out"]) if t else None,
            _fp(t["dqda"]) if t else None, _fp(t["grads_c"]) if t else None, _fp(t["grads_a"]) if t else None)
        return t
