"""Python face of a NAF population handle (rlc_naf_* in include/rlcontrol_hip.h)."""
import ctypes
from collections import OrderedDict

import numpy as np

from . import _lib
from ._lib import check, dptr, f64, fptr, iptr
from .hip_pop import Population


NORM_TYPES = {"none": 0, "input_norm": 0, "layer": 1}


def param_layout(S, A, L1, L2, norm_type="input_norm"):
    """name -> (offset, shape), variable creation order of naf_network.py:79-107; with norm_type 'layer' the trunk and
    both branch hidden layers are followed by their layer-norm beta and gamma (tf.contrib.layers.layer_norm creates beta
    first; base_network.py:53-56, naf_network.py:83,87,93)."""
    if norm_type not in NORM_TYPES:
        raise ValueError("norm_type %r is not implemented (implemented: %s)" % (norm_type, ", ".join(sorted(NORM_TYPES))))
    ln = lambda tag, n: [("L%sb" % tag, (n,)), ("L%sg" % tag, (n,))] if NORM_TYPES[norm_type] == 1 else []
    segs = [("W1", (S, L1)), ("b1", (L1,))] + ln("1", L1) + [("Wa2", (L1, L2)), ("ba2", (L2,))] + ln("a2", L2) + \
           [("Wa3", (L2, A)), ("ba3", (A,)), ("Wv2", (L1, L2)), ("bv2", (L2,))] + ln("v2", L2) + \
           [("Wv3", (L2, 1)), ("bv3", (1,))]
    for c in range(A):
        segs += [("Wd%d" % c, (L1, 1)), ("bd%d" % c, (1,))]
    for c in range(A - 1):
        segs += [("Wn%d" % c, (L1, A - 1 - c)), ("bn%d" % c, (A - 1 - c,))]
    out, p = OrderedDict(), 0
    for name, shp in segs:
        out[name] = (p, shp)
        p += int(np.prod(shp))
    return out, p


def init_params(S, A, L1, L2, seed, norm_type="input_norm"):
    """fully_connected defaults (naf_network.py:81-107): Glorot-uniform weights, zero biases; V output weights
    U(+-3e-3) (:96); layer-norm beta zeros, gamma ones.  numpy RandomState(seed) stands in for TF's stream
    (distribution parity only)."""
    rng = np.random.RandomState(seed)
    lay, P = param_layout(S, A, L1, L2, norm_type)
    th = np.zeros(P, np.float32)
    for name, (off, shp) in lay.items():
        if name.startswith("b"):
            continue
        if name.startswith("L"):
            if name.endswith("g"):
                th[off:off + shp[0]] = 1.0
            continue
        n = int(np.prod(shp))
        lim = 3e-3 if name == "Wv3" else np.sqrt(6.0 / (shp[0] + shp[1]))
        th[off:off + n] = rng.uniform(-lim, lim, n)
    return th


class NAFPopulation(Population):
    BLOB = {"theta": 0, "theta_target": 1, "adam_m": 2, "adam_v": 3}
    TAP = {"q": 0, "y": 1, "V": 2, "grads": 3}

    def __init__(self, n_agents, state_dim, action_dim, l1_dim, l2_dim, batch_size, buffer_size, tau, state_min,
                 state_max, action_max, learning_rate, seeds, clip_state=True, device=0, norm_type="input_norm",
                 action_min=None):
        self._init_base(n_agents, state_dim, action_dim, batch_size)
        self.dims = (self.S, self.A, int(l1_dim), int(l2_dim))
        self.norm_type = norm_type
        self.layout, self.P = param_layout(*self.dims, norm_type=norm_type)
        bc = lambda v, n: np.ascontiguousarray(np.broadcast_to(np.asarray(v, np.float32).reshape(-1), (n,)))
        self._keep = dict(smin=bc(state_min, self.S), smax=bc(state_max, self.S), amax=bc(action_max, self.A),
                          lr=bc(learning_rate, self.n_agents),
                          seed=np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, np.uint64).reshape(-1), (self.n_agents,))))
        cfg = _lib.rlc_naf_config()
        cfg.device, cfg.n_agents, cfg.state_dim, cfg.action_dim = int(device), self.n_agents, self.S, self.A
        cfg.l1_dim, cfg.l2_dim = self.dims[2:]
        cfg.batch_size, cfg.clip_state, cfg.buffer_size, cfg.tau = self.B, 1 if clip_state else 0, int(buffer_size), float(tau)
        cfg.norm_type = NORM_TYPES[norm_type]
        cfg.state_min, cfg.state_max, cfg.action_max = fptr(self._keep["smin"]), fptr(self._keep["smax"]), fptr(self._keep["amax"])
        # lower clip of the exploration draw in the on-device loop (naf_network.py:176); default: a symmetric box
        self._keep["amin"] = bc(-np.asarray(self._keep["amax"]) if action_min is None else action_min, self.A)
        cfg.action_min = fptr(self._keep["amin"])
        cfg.learning_rate = fptr(self._keep["lr"])
        cfg.seed = self._keep["seed"].ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
        check(self._lib.rlc_naf_create(ctypes.byref(cfg), ctypes.byref(self._h)))

    def set_blob(self, agent, which, values):
        v = np.ascontiguousarray(values, np.float32).reshape(-1)
        check(self._lib.rlc_naf_set_blob(self._h, int(agent), self.BLOB[which], fptr(v), ctypes.c_int64(v.size)))

    def get_blob(self, agent, which):
        out = np.empty(self.P, np.float32)
        check(self._lib.rlc_naf_get_blob(self._h, int(agent), self.BLOB[which], fptr(out), ctypes.c_int64(self.P)))
        return out

    def set_params(self, agent, theta, init_target=True):
        self.set_blob(agent, "theta", theta)
        if init_target:
            check(self._lib.rlc_naf_init_target(self._h, int(agent)))

    def get_beta_powers(self, agent):
        out = np.empty(2, np.float32)
        check(self._lib.rlc_naf_get_beta_powers(self._h, int(agent), fptr(out)))
        return out

    def act(self, states, first_agent=0, with_lcols=False):
        s = f64(states).reshape(-1, self.S)
        mu = np.empty((s.shape[0], self.A), np.float32)
        lc = np.empty((s.shape[0], self.A * (self.A + 1) // 2), np.float32) if with_lcols else None
        check(self._lib.rlc_naf_act(self._h, int(first_agent), ctypes.c_int32(s.shape[0]), dptr(s), fptr(mu),
                                    fptr(lc) if lc is not None else None))
        return (mu, lc) if with_lcols else mu

    def act_queue(self, states, first_agent=0):
        """queue the greedy forward (mu and the L columns) for `states` behind the work already on the handle's stream"""
        s = f64(states).reshape(-1, self.S)
        check(self._lib.rlc_naf_act_queue(self._h, int(first_agent), ctypes.c_int32(s.shape[0]), dptr(s)))
        return s.shape[0]

    def act_fetch(self, n, first_agent=0):
        mu = np.empty((int(n), self.A), np.float32)
        lc = np.empty((int(n), self.A * (self.A + 1) // 2), np.float32)
        check(self._lib.rlc_naf_act_fetch(self._h, int(first_agent), ctypes.c_int32(int(n)), fptr(mu), fptr(lc)))
        return mu, lc

    def update(self, n_updates=1, host_indices=None):
        idx = None
        if host_indices is not None:
            idx = np.ascontiguousarray(host_indices, np.int64)
            if idx.size != self.n_agents * int(n_updates) * self.B:
                raise ValueError("host_indices must hold n_agents*n_updates*batch_size entries")
        check(self._lib.rlc_naf_update(self._h, ctypes.c_int32(int(n_updates)), iptr(idx) if idx is not None else None))

    def update_batch(self, agent, states, actions, next_states, rewards, gammas):
        r = f64(rewards).reshape(-1)
        n = r.size
        s, s2 = f64(states).reshape(n, self.S), f64(next_states).reshape(n, self.S)
        a, g = f64(actions).reshape(n, self.A), f64(gammas).reshape(n)
        check(self._lib.rlc_naf_update_batch(self._h, int(agent), ctypes.c_int32(n), dptr(s), dptr(a), dptr(s2), dptr(r),
                                             dptr(g)))

    KERNEL = {"auto": 0, "generic": 1, "mfma": 2}

    def set_kernel(self, name):
        check(self._lib.rlc_naf_set_kernel(self._h, self.KERNEL[name]))

    def kernel_in_use(self):
        out = ctypes.c_int32(0)
        check(self._lib.rlc_naf_get_kernel(self._h, ctypes.byref(out)))
        return {v: k for k, v in self.KERNEL.items()}[out.value]

    def enable_grad_taps(self, on=True):
        check(self._lib.rlc_naf_enable_grad_taps(self._h, 1 if on else 0))

    def last_tap(self, agent, which):
        n = {"q": self.B, "y": self.B, "V": self.B, "grads": self.P}[which]
        out = np.empty(n, np.float32)
        check(self._lib.rlc_naf_last_tap(self._h, int(agent), self.TAP[which], fptr(out), ctypes.c_int64(n)))
        return out
