"""Python face of one ``rlc_ddpg`` handle: a population of independent DDPG agents on one MI355X.

Thin, typed wrappers over the C ABI (include/rlcontrol_hip.h); no arithmetic happens here.
``n_agents == 1`` is the reference's one-agent-per-process deployment (main.py:111-185); larger
populations are the reference's INDEX sweep (settings x seeds) co-resident on one GPU.
"""
import ctypes
from collections import OrderedDict

import numpy as np

from . import _lib
from ._lib import check, dptr, f64, fptr, iptr
from .hip_pop import Population


NORM_TYPES = {"none": 0, "input_norm": 0, "layer": 1}


def param_layout(S, A, H1, HA, HC, norm_type="input_norm", separate_networks=False):
    """name -> (offset, shape), variable creation order of hydra_ddpg_network.py:100-140; with norm_type 'layer' every
    hidden layer is followed by its LayerNorm beta then gamma (base_network.py:53-56), with separate networks
    (actor_network.py:73-96, critic_network.py:77-99) the critic block starts with a first layer of its own."""
    ln = lambda tag, n: [(tag + "b", (n,)), (tag + "g", (n,))] if NORM_TYPES[norm_type] else []
    items = [("W1", (S, H1)), ("b1", (H1,))] + ln("l1", H1) + [("Wa2", (H1, HA)), ("ba2", (HA,))] + ln("l2", HA) + \
            [("Wa3", (HA, A)), ("ba3", (A,))]
    if separate_networks:
        items += [("Wc1", (S, H1)), ("bc1", (H1,))] + ln("lc", H1)
    items += [("Wc2", (H1 + A, HC)), ("bc2", (HC,))] + ln("l3", HC) + [("Wc3", (HC, 1)), ("bc3", (1,))]
    out = OrderedDict()
    p = 0
    for name, shp in items:
        out[name] = (p, shp)
        p += int(np.prod(shp))
    return out, p


def init_params(S, A, H1, HA, HC, seed, norm_type="input_norm", separate_networks=False):
    """Initial weights with the reference's initialiser families (hydra_ddpg_network.py:101-140):
    hidden W and b ~ U(+-sqrt(3/fan_in)) (variance_scaling_initializer(factor=1, FAN_IN, uniform);
    for a 1-D bias [n] TF takes fan_in = n), output-layer W, b ~ U(+-3e-3); LayerNorm beta 0, gamma 1.  TensorFlow's own
    Philox stream under tf.set_random_seed(seed) (agents/DDPG.py:21) cannot be reproduced without
    TensorFlow, so the draws come from numpy RandomState(seed): same distributions, different
    numbers ("parity unpinned")."""
    rng = np.random.RandomState(seed)
    lay, P = param_layout(S, A, H1, HA, HC, norm_type, separate_networks)
    theta = np.zeros(P, np.float32)
    for name, (off, shp) in lay.items():
        n = int(np.prod(shp))
        if name[0] == "l":
            theta[off:off + n] = 1.0 if name.endswith("g") else 0.0
            continue
        lim = 3e-3 if name in ("Wa3", "ba3", "Wc3", "bc3") else np.sqrt(3.0 / shp[0])
        theta[off:off + n] = rng.uniform(-lim, lim, n).astype(np.float32)
    return theta


class DDPGPopulation(Population):
    BLOB = {"theta": 0, "theta_target": 1, "actor_m": 2, "actor_v": 3, "critic_m": 4, "critic_v": 5}
    TAP = {"q": 0, "y": 1, "a_out": 2, "dqda": 3, "grads_c": 4, "grads_a": 5}
    KERNEL = {"auto": 0, "generic": 1, "mfma": 2}

    def __init__(self, n_agents, state_dim, action_dim, shared_l1_dim, actor_l2_dim, critic_l2_dim, batch_size,
                 buffer_size, tau, state_min, state_max, action_min, action_max, actor_lr, critic_lr, seeds,
                 clip_state=True, ou_theta=0.15, ou_mu=0.0, ou_sigma=0.2, device=0, norm_type="input_norm",
                 separate_networks=False):
        self._init_base(n_agents, state_dim, action_dim, batch_size)
        self.H1, self.HA, self.HC = int(shared_l1_dim), int(actor_l2_dim), int(critic_l2_dim)
        if norm_type not in NORM_TYPES:
            raise ValueError("norm_type %r: the HIP path implements 'none', 'input_norm' and 'layer' (the reference's "
                             "'batch', base_network.py:57-59, is not implemented)" % (norm_type,))
        self.norm_type, self.separate_networks = norm_type, bool(separate_networks)
        self.layout, self.P = param_layout(self.S, self.A, self.H1, self.HA, self.HC, norm_type, separate_networks)
        bc = lambda v, n: np.ascontiguousarray(np.broadcast_to(np.asarray(v, np.float32).reshape(-1), (n,)))
        self._keep = dict(
            smin=bc(state_min, self.S), smax=bc(state_max, self.S), amin=bc(action_min, self.A),
            amax=bc(action_max, self.A), lra=bc(actor_lr, self.n_agents), lrc=bc(critic_lr, self.n_agents),
            seed=np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, np.uint64).reshape(-1), (self.n_agents,))))
        cfg = _lib.rlc_ddpg_config()
        cfg.device = int(device)
        cfg.n_agents = self.n_agents
        cfg.state_dim, cfg.action_dim = self.S, self.A
        cfg.shared_l1_dim, cfg.actor_l2_dim, cfg.critic_l2_dim = self.H1, self.HA, self.HC
        cfg.batch_size = self.B
        cfg.buffer_size = int(buffer_size)
        cfg.clip_state = 1 if clip_state else 0
        cfg.norm_type = NORM_TYPES[norm_type]
        cfg.separate_networks = 1 if separate_networks else 0
        cfg.tau = float(tau)
        cfg.state_min, cfg.state_max = fptr(self._keep["smin"]), fptr(self._keep["smax"])
        cfg.action_min, cfg.action_max = fptr(self._keep["amin"]), fptr(self._keep["amax"])
        cfg.actor_lr, cfg.critic_lr = fptr(self._keep["lra"]), fptr(self._keep["lrc"])
        cfg.seed = self._keep["seed"].ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
        cfg.ou_theta, cfg.ou_mu, cfg.ou_sigma = float(ou_theta), float(ou_mu), float(ou_sigma)
        check(self._lib.rlc_ddpg_create(ctypes.byref(cfg), ctypes.byref(self._h)))

    # ---- parameters -----------------------------------------------------------------------
    def set_blob(self, agent, which, values):
        v = np.ascontiguousarray(values, np.float32).reshape(-1)
        check(self._lib.rlc_ddpg_set_blob(self._h, int(agent), self.BLOB[which], fptr(v), ctypes.c_int64(v.size)))

    def get_blob(self, agent, which):
        out = np.empty(self.P, np.float32)
        check(self._lib.rlc_ddpg_get_blob(self._h, int(agent), self.BLOB[which], fptr(out), ctypes.c_int64(self.P)))
        return out

    def set_params(self, agent, theta, init_target=True):
        self.set_blob(agent, "theta", theta)
        if init_target:
            check(self._lib.rlc_ddpg_init_target(self._h, int(agent)))

    def get_beta_powers(self, agent):
        out = np.empty(4, np.float32)
        check(self._lib.rlc_ddpg_get_beta_powers(self._h, int(agent), fptr(out)))
        return out

    def set_beta_powers(self, agent, pw4):
        v = np.ascontiguousarray(pw4, np.float32).reshape(4)
        check(self._lib.rlc_ddpg_set_beta_powers(self._h, int(agent), fptr(v)))

    def named(self, blob):
        return OrderedDict((k, blob[o:o + int(np.prod(s))].reshape(s)) for k, (o, s) in self.layout.items())

    # ---- acting ---------------------------------------------------------------------------
    def act(self, states, first_agent=0, explore=False):
        s = f64(states).reshape(-1, self.S)
        out = np.empty((s.shape[0], self.A), np.float32)
        fn = self._lib.rlc_ddpg_act_explore if explore else self._lib.rlc_ddpg_act
        check(fn(self._h, int(first_agent), ctypes.c_int32(s.shape[0]), dptr(s), fptr(out)))
        return out

    def act_queue(self, states, first_agent=0):
        """queue the greedy forward for `states` behind the work already on the handle's stream (no synchronisation)"""
        s = f64(states).reshape(-1, self.S)
        check(self._lib.rlc_ddpg_act_queue(self._h, int(first_agent), ctypes.c_int32(s.shape[0]), dptr(s)))
        return s.shape[0]

    def act_fetch(self, n, first_agent=0):
        out = np.empty((int(n), self.A), np.float32)
        check(self._lib.rlc_ddpg_act_fetch(self._h, int(first_agent), ctypes.c_int32(int(n)), fptr(out)))
        return out

    def reset_noise(self, first_agent=0, n=None):
        check(self._lib.rlc_ddpg_reset_noise(self._h, int(first_agent), int(self.n_agents if n is None else n)))

    def qval(self, agent, states, actions):
        s = f64(states).reshape(-1, self.S)
        a = f64(actions).reshape(-1, self.A)
        out = np.empty(s.shape[0], np.float32)
        check(self._lib.rlc_ddpg_qval(self._h, int(agent), ctypes.c_int32(s.shape[0]), dptr(s), dptr(a), fptr(out)))
        return out

    # ---- learning -------------------------------------------------------------------------
    def update(self, n_updates=1, host_indices=None):
        if host_indices is None:
            check(self._lib.rlc_ddpg_update(self._h, ctypes.c_int32(int(n_updates)), None))
            return
        idx = np.ascontiguousarray(host_indices, np.int64)
        if idx.size != self.n_agents * int(n_updates) * self.B:
            raise ValueError("host_indices must hold n_agents*n_updates*batch_size entries")
        check(self._lib.rlc_ddpg_update(self._h, ctypes.c_int32(int(n_updates)), iptr(idx)))

    def update_batch(self, agent, states, actions, next_states, rewards, gammas):
        r = f64(rewards).reshape(-1)
        n = r.size
        s, s2 = f64(states).reshape(n, self.S), f64(next_states).reshape(n, self.S)
        a, g = f64(actions).reshape(n, self.A), f64(gammas).reshape(n)
        check(self._lib.rlc_ddpg_update_batch(self._h, int(agent), ctypes.c_int32(n), dptr(s), dptr(a), dptr(s2),
                                              dptr(r), dptr(g)))

    def set_kernel(self, name):
        check(self._lib.rlc_ddpg_set_kernel(self._h, self.KERNEL[name]))

    def set_split(self, n_workgroups):
        """latency mode: every agent's minibatch over n_workgroups CUs (1 = off); MFMA shapes only"""
        check(self._lib.rlc_ddpg_set_split(self._h, ctypes.c_int32(int(n_workgroups))))

    def debug_fail_next_split(self):
        """test hook: the next latency-mode launch finds its barrier error word set (include/rlcontrol_hip.h)"""
        check(self._lib.rlc_debug_fail_next_split(self._h))

    def kernel_in_use(self):
        out = ctypes.c_int32(0)
        check(self._lib.rlc_ddpg_get_kernel(self._h, ctypes.byref(out)))
        return {v: k for k, v in self.KERNEL.items()}[out.value]

    def enable_grad_taps(self, on=True):
        check(self._lib.rlc_ddpg_enable_grad_taps(self._h, 1 if on else 0))

    def last_tap(self, agent, which):
        n = {"q": self.B, "y": self.B, "a_out": self.B * self.A, "dqda": self.B * self.A,
             "grads_c": self.P, "grads_a": self.P}[which]
        out = np.empty(n, np.float32)
        check(self._lib.rlc_ddpg_last_tap(self._h, int(agent), self.TAP[which], fptr(out), ctypes.c_int64(n)))
        return out
