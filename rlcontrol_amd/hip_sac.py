"""Python face of a SoftActorCritic (SAC-v1) population handle (rlc_sac_* in include/rlcontrol_hip.h)."""
import ctypes
from collections import OrderedDict

import numpy as np

from . import _lib
from ._lib import check, dptr, f64, fptr, iptr
from .hip_pop import Population


NORM_TYPES = {"none": 0, "input_norm": 0, "layer": 1}


def param_layout(S, A, L1A, L2A, L1C, L2C, norm_type="input_norm"):
    """name -> (offset, shape), variable creation order under 'main' (sac_network.py:152-172); with norm_type 'layer'
    every hidden layer is followed by its layer-norm beta and gamma (tf.contrib.layers.layer_norm creates beta first)."""
    if norm_type not in NORM_TYPES:
        raise ValueError("norm_type %r is not implemented (implemented: %s)" % (norm_type, ", ".join(sorted(NORM_TYPES))))
    ln = NORM_TYPES[norm_type] == 1
    spec = []
    for pre, w1, w2, l1, l2 in (("p", (S, L1A), (L1A, L2A), L1A, L2A), ("q", (S, L1C), (L1C + A, L2C), L1C, L2C),
                                ("v", (S, L1C), (L1C, L2C), L1C, L2C)):
        spec += [(pre + "W1", w1), (pre + "b1", (l1,))]
        if ln:
            spec += [(pre + "L1b", (l1,)), (pre + "L1g", (l1,))]
        spec += [(pre + "W2", w2), (pre + "b2", (l2,))]
        if ln:
            spec += [(pre + "L2b", (l2,)), (pre + "L2g", (l2,))]
        if pre == "p":
            spec += [("pWm", (L2A, A)), ("pbm", (A,)), ("pWs", (L2A, A)), ("pbs", (A,))]
        else:
            spec += [(pre + "W3", (l2, 1)), (pre + "b3", (1,))]
    out, p = OrderedDict(), 0
    for name, shp in spec:
        out[name] = (p, shp)
        p += int(np.prod(shp))
    return out, p


def init_params(S, A, L1A, L2A, L1C, L2C, seed, norm_type="input_norm"):
    """Initialiser families of sac_network.py: hidden layers and the mu head U(+-sqrt(3/fan_in)) (:178-270),
    log_std head W ~ U(0,1), b ~ U(+-3e-3) (:273-280), Q / V output layers U(+-3e-3) (:198-201,227-230);
    numpy RandomState(seed) instead of TF's unreproducible stream (distribution parity only)."""
    rng = np.random.RandomState(seed)
    lay, P = param_layout(S, A, L1A, L2A, L1C, L2C, norm_type)
    th = np.zeros(P, np.float32)
    for name, (off, shp) in lay.items():
        n = int(np.prod(shp))
        if name[1] == "L":                 # layer norm: beta zeros, gamma ones (tf.contrib.layers.layer_norm defaults)
            th[off:off + n] = 1.0 if name.endswith("g") else 0.0
        elif name == "pWs":
            th[off:off + n] = rng.uniform(0.0, 1.0, n)
        elif name in ("pbs", "qW3", "qb3", "vW3", "vb3"):
            th[off:off + n] = rng.uniform(-3e-3, 3e-3, n)
        else:
            lim = np.sqrt(3.0 / shp[0])
            th[off:off + n] = rng.uniform(-lim, lim, n)
    return th


class SACPopulation(Population):
    BLOB = {"theta": 0, "theta_target": 1, "adam_m": 2, "adam_v": 3}
    TAP = {"q": 0, "v": 1, "logp": 2, "q_pi": 3, "loss": 4, "grads": 5}

    def __init__(self, n_agents, state_dim, action_dim, actor_l1_dim, actor_l2_dim, critic_l1_dim, critic_l2_dim,
                 batch_size, buffer_size, tau, state_min0, state_max0, action_max0, pi_lr, qf_vf_lr, entropy_scale,
                 seeds, clip_state=True, device=0, norm_type="input_norm"):
        self._init_base(n_agents, state_dim, action_dim, batch_size)
        self.dims = (self.S, self.A, int(actor_l1_dim), int(actor_l2_dim), int(critic_l1_dim), int(critic_l2_dim))
        self.norm_type = norm_type
        self.layout, self.P = param_layout(*self.dims, norm_type=norm_type)
        bc = lambda v: np.ascontiguousarray(np.broadcast_to(np.asarray(v, np.float32).reshape(-1), (self.n_agents,)))
        self._keep = dict(lp=bc(pi_lr), lq=bc(qf_vf_lr), al=bc(entropy_scale),
                          seed=np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, np.uint64).reshape(-1), (self.n_agents,))))
        cfg = _lib.rlc_sac_config()
        cfg.device, cfg.n_agents, cfg.state_dim, cfg.action_dim = int(device), self.n_agents, self.S, self.A
        cfg.actor_l1_dim, cfg.actor_l2_dim, cfg.critic_l1_dim, cfg.critic_l2_dim = self.dims[2:]
        cfg.batch_size, cfg.clip_state, cfg.buffer_size = self.B, 1 if clip_state else 0, int(buffer_size)
        cfg.tau, cfg.state_min0, cfg.state_max0, cfg.action_max0 = float(tau), float(state_min0), float(state_max0), float(action_max0)
        cfg.pi_lr, cfg.qf_vf_lr, cfg.entropy_scale = fptr(self._keep["lp"]), fptr(self._keep["lq"]), fptr(self._keep["al"])
        cfg.seed = self._keep["seed"].ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
        cfg.norm_type = NORM_TYPES[norm_type]
        check(self._lib.rlc_sac_create(ctypes.byref(cfg), ctypes.byref(self._h)))

    def set_blob(self, agent, which, values):
        v = np.ascontiguousarray(values, np.float32).reshape(-1)
        check(self._lib.rlc_sac_set_blob(self._h, int(agent), self.BLOB[which], fptr(v), ctypes.c_int64(v.size)))

    def get_blob(self, agent, which):
        out = np.empty(self.P, np.float32)
        check(self._lib.rlc_sac_get_blob(self._h, int(agent), self.BLOB[which], fptr(out), ctypes.c_int64(self.P)))
        return out

    def set_params(self, agent, theta, init_target=True):
        self.set_blob(agent, "theta", theta)
        if init_target:
            check(self._lib.rlc_sac_init_target(self._h, int(agent)))

    def get_beta_powers(self, agent):
        out = np.empty(4, np.float32)
        check(self._lib.rlc_sac_get_beta_powers(self._h, int(agent), fptr(out)))
        return out

    def act(self, states, first_agent=0, sample=False, eps=None):
        s = f64(states).reshape(-1, self.S)
        out = np.empty((s.shape[0], self.A), np.float32)
        e = None if eps is None else np.ascontiguousarray(eps, np.float32).reshape(s.shape[0], self.A)
        check(self._lib.rlc_sac_act(self._h, int(first_agent), ctypes.c_int32(s.shape[0]), dptr(s),
                                    ctypes.c_int32(1 if sample else 0), fptr(e) if e is not None else None, fptr(out)))
        return out

    def act_queue(self, states, first_agent=0, sample=False, eps=None):
        """queue the acting forward for `states` behind the work already on the handle's stream (no synchronisation)"""
        s = f64(states).reshape(-1, self.S)
        e = None if eps is None else np.ascontiguousarray(eps, np.float32).reshape(s.shape[0], self.A)
        check(self._lib.rlc_sac_act_queue(self._h, int(first_agent), ctypes.c_int32(s.shape[0]), dptr(s),
                                          ctypes.c_int32(1 if sample else 0), fptr(e) if e is not None else None))
        return s.shape[0]

    def act_fetch(self, n, first_agent=0):
        out = np.empty((int(n), self.A), np.float32)
        check(self._lib.rlc_sac_act_fetch(self._h, int(first_agent), ctypes.c_int32(int(n)), fptr(out)))
        return out

    def update(self, n_updates=1, host_indices=None, eps=None):
        idx = None
        if host_indices is not None:
            idx = np.ascontiguousarray(host_indices, np.int64)
            if idx.size != self.n_agents * int(n_updates) * self.B:
                raise ValueError("host_indices must hold n_agents*n_updates*batch_size entries")
        e = None
        if eps is not None:
            e = np.ascontiguousarray(eps, np.float32)
            if e.size != self.n_agents * int(n_updates) * self.B * self.A:
                raise ValueError("eps must hold n_agents*n_updates*batch_size*action_dim entries")
        check(self._lib.rlc_sac_update(self._h, ctypes.c_int32(int(n_updates)), iptr(idx) if idx is not None else None,
                                       fptr(e) if e is not None else None))

    def update_batch(self, agent, states, actions, next_states, rewards, gammas, eps=None):
        r = f64(rewards).reshape(-1)
        n = r.size
        s, s2 = f64(states).reshape(n, self.S), f64(next_states).reshape(n, self.S)
        a, g = f64(actions).reshape(n, self.A), f64(gammas).reshape(n)
        e = None if eps is None else np.ascontiguousarray(eps, np.float32).reshape(n, self.A)
        check(self._lib.rlc_sac_update_batch(self._h, int(agent), ctypes.c_int32(n), dptr(s), dptr(a), dptr(s2), dptr(r),
                                             dptr(g), fptr(e) if e is not None else None))

    KERNEL = {"auto": 0, "generic": 1, "mfma": 2}

    def set_kernel(self, name):
        check(self._lib.rlc_sac_set_kernel(self._h, self.KERNEL[name]))

    def kernel_in_use(self):
        out = ctypes.c_int32(0)
        check(self._lib.rlc_sac_get_kernel(self._h, ctypes.byref(out)))
        return {v: k for k, v in self.KERNEL.items()}[out.value]

    def enable_grad_taps(self, on=True):
        check(self._lib.rlc_sac_enable_grad_taps(self._h, 1 if on else 0))

    def last_tap(self, agent, which):
        n = {"q": self.B, "v": self.B, "logp": self.B, "q_pi": self.B, "loss": 3, "grads": self.P}[which]
        out = np.empty(n, np.float32)
        check(self._lib.rlc_sac_last_tap(self._h, int(agent), self.TAP[which], fptr(out), ctypes.c_int64(n)))
        return out
