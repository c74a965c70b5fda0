"""Python face of a ReverseKL / ForwardKL population handle (rlc_kl_* in include/rlcontrol_hip.h)."""
import ctypes
import math
from collections import OrderedDict

import numpy as np

from . import _lib
from ._lib import check, dptr, f64, fptr, iptr
from .hip_pop import Population
from .utils.quadrature import interior_action_nodes, sparse_grid_action_nodes

KINDS = {"reverse": 1, "forward": 2}
OPTIM_TYPES = {"intg": 0, "hard_intg": 1, "ll": 2, "hard_ll": 3}
Q_UPDATE_TYPES = {"non_sac": 0, "sac": 1}


def param_layout(S, A, L1A, L2A, L1C, L2C):
    """name -> (offset, shape); modules pi_net, q_net, v_net (reversekl_network.py:47-50), weights [in, out]:
    the transpose of the nn.Linear.weight tensors of a state_dict."""
    out, p = OrderedDict(), 0
    for name, shp in (("pW1", (S, L1A)), ("pb1", (L1A,)), ("pW2", (L1A, L2A)), ("pb2", (L2A,)),
                      ("pWm", (L2A, A)), ("pbm", (A,)), ("pWs", (L2A, A)), ("pbs", (A,)),
                      ("qW1", (S + A, L1C)), ("qb1", (L1C,)), ("qW2", (L1C, L2C)), ("qb2", (L2C,)),
                      ("qW3", (L2C, 1)), ("qb3", (1,)),
                      ("vW1", (S, L1C)), ("vb1", (L1C,)), ("vW2", (L1C, L2C)), ("vb2", (L2C,)),
                      ("vW3", (L2C, 1)), ("vb3", (1,))):
        out[name] = (p, shp)
        p += int(np.prod(shp))
    return out, p


def init_params(S, A, L1A, L2A, L1C, L2C, seed):
    """nn.Linear's default initialiser for the hidden layers (W and b ~ U(+-1/sqrt(fan_in))) and U(+-3e-3) for the
    mean / log_std / Q / V output layers (reversekl_network.py:246-247,265-266,290-296); numpy RandomState(seed)
    instead of torch's generator (distribution parity only)."""
    rng = np.random.RandomState(seed)
    lay, P = param_layout(S, A, L1A, L2A, L1C, L2C)
    th = np.zeros(P, np.float32)
    fan_in = {}
    for name, (off, shp) in lay.items():
        n = int(np.prod(shp))
        layer = name[0] + name[2:]
        if name[1] == "W":
            fan_in[layer] = shp[0]
        lim = 3e-3 if name[2:] in ("m", "s", "3") else 1.0 / math.sqrt(fan_in[layer])
        th[off:off + n] = rng.uniform(-lim, lim, n)
    return th


class KLPopulation(Population):
    BLOB = {"theta": 0, "theta_target": 1, "adam_m": 2, "adam_v": 3}
    TAP = {"q": 0, "v": 1, "logp": 2, "q_pi": 3, "loss": 4, "grads": 5, "intgrl_q": 6}

    def __init__(self, kind, n_agents, state_dim, action_dim, actor_l1_dim, actor_l2_dim, critic_l1_dim, critic_l2_dim,
                 batch_size, buffer_size, tau, action_max0, pi_lr, qf_vf_lr, entropy_scale, seeds, n_param,
                 optim_type="intg", q_update_type="non_sac", device=0, nodes=None, l_param=None, action_max=None):
        if kind not in KINDS:
            raise ValueError("kind must be 'reverse' or 'forward'")
        if optim_type not in OPTIM_TYPES:
            raise ValueError("invalid config.optim_type %r" % (optim_type,))
        if q_update_type not in Q_UPDATE_TYPES:
            raise ValueError("invalid config.q_update_type")        # reversekl_network.py:160
        self._init_base(n_agents, state_dim, action_dim, batch_size)
        self.kind, self.optim_type, self.q_update_type = kind, optim_type, q_update_type
        self.dims = (self.S, self.A, int(actor_l1_dim), int(actor_l2_dim), int(critic_l1_dim), int(critic_l2_dim))
        self.layout, self.P = param_layout(*self.dims)
        if nodes is None:
            # one action dimension: the Clenshaw-Curtis line rule on N_param points; above it the sparse grid of level
            # l_param, its nodes scaled by the whole action_max vector (reversekl_network.py:64-108) while the policy
            # itself is scaled by action_max[0] (:47)
            if self.A == 1:
                nodes = interior_action_nodes(int(n_param), float(action_max0))
            else:
                if l_param is None:
                    raise ValueError("action_dim > 1 needs l_param (the sparse grid's level)")
                nodes = sparse_grid_action_nodes(int(l_param), self.A, action_max0 if action_max is None else action_max)
        node_a = np.ascontiguousarray(nodes[0], np.float32).reshape(-1)
        node_w = np.ascontiguousarray(nodes[1], np.float32).reshape(-1)
        if node_a.size != node_w.size * self.A:
            raise ValueError("nodes: actions [K, action_dim] and weights [K] differ in length")
        self.n_nodes = int(node_w.size)
        bc = lambda v: np.ascontiguousarray(np.broadcast_to(np.asarray(v, np.float32).reshape(-1), (self.n_agents,)))
        self._keep = dict(lp=bc(pi_lr), lq=bc(qf_vf_lr), al=bc(entropy_scale), na=node_a, nw=node_w,
                          seed=np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, np.uint64).reshape(-1), (self.n_agents,))))
        cfg = _lib.rlc_kl_config()
        cfg.device, cfg.n_agents, cfg.state_dim, cfg.action_dim = int(device), self.n_agents, self.S, self.A
        cfg.actor_l1_dim, cfg.actor_l2_dim, cfg.critic_l1_dim, cfg.critic_l2_dim = self.dims[2:]
        cfg.batch_size, cfg.buffer_size = self.B, int(buffer_size)
        cfg.kind, cfg.optim_type, cfg.q_update_type = KINDS[kind], OPTIM_TYPES[optim_type], Q_UPDATE_TYPES[q_update_type]
        cfg.n_nodes = self.n_nodes
        cfg.tau, cfg.action_max0 = float(tau), float(action_max0)
        cfg.node_actions, cfg.node_weights = fptr(node_a), fptr(node_w)
        cfg.pi_lr, cfg.qf_vf_lr, cfg.entropy_scale = fptr(self._keep["lp"]), fptr(self._keep["lq"]), fptr(self._keep["al"])
        cfg.seed = self._keep["seed"].ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
        check(self._lib.rlc_kl_create(ctypes.byref(cfg), ctypes.byref(self._h)))

    def set_blob(self, agent, which, values):
        v = np.ascontiguousarray(values, np.float32).reshape(-1)
        check(self._lib.rlc_kl_set_blob(self._h, int(agent), self.BLOB[which], fptr(v), ctypes.c_int64(v.size)))

    def get_blob(self, agent, which):
        out = np.empty(self.P, np.float32)
        check(self._lib.rlc_kl_get_blob(self._h, int(agent), self.BLOB[which], fptr(out), ctypes.c_int64(self.P)))
        return out

    def set_params(self, agent, theta, init_target=True):
        self.set_blob(agent, "theta", theta)
        if init_target:
            check(self._lib.rlc_kl_init_target(self._h, int(agent)))

    def debug_fail_next_split(self):
        check(self._lib.rlc_debug_fail_next_split(self._h))

    def set_split(self, n_workgroups):
        """latency mode: the node passes of each agent's action integral over that many CUs (1 = off)"""
        check(self._lib.rlc_kl_set_split(self._h, ctypes.c_int32(int(n_workgroups))))

    def get_step(self, agent):
        out = ctypes.c_int32(0)
        check(self._lib.rlc_kl_get_step(self._h, int(agent), ctypes.byref(out)))
        return out.value

    def set_step(self, agent, step):
        check(self._lib.rlc_kl_set_step(self._h, int(agent), ctypes.c_int32(int(step))))

    def act(self, states, first_agent=0, sample=False, eps=None):
        s = f64(states).reshape(-1, self.S)
        out = np.empty((s.shape[0], self.A), np.float32)
        e = None if eps is None else np.ascontiguousarray(eps, np.float32).reshape(s.shape[0], self.A)
        check(self._lib.rlc_kl_act(self._h, int(first_agent), ctypes.c_int32(s.shape[0]), dptr(s),
                                   ctypes.c_int32(1 if sample else 0), fptr(e) if e is not None else None, fptr(out)))
        return out

    def act_queue(self, states, first_agent=0, sample=False, eps=None):
        """queue the acting forward for `states` behind the work already on the handle's stream (no synchronisation)"""
        s = f64(states).reshape(-1, self.S)
        e = None if eps is None else np.ascontiguousarray(eps, np.float32).reshape(s.shape[0], self.A)
        check(self._lib.rlc_kl_act_queue(self._h, int(first_agent), ctypes.c_int32(s.shape[0]), dptr(s),
                                          ctypes.c_int32(1 if sample else 0), fptr(e) if e is not None else None))
        return s.shape[0]

    def act_fetch(self, n, first_agent=0):
        out = np.empty((int(n), self.A), np.float32)
        check(self._lib.rlc_kl_act_fetch(self._h, int(first_agent), ctypes.c_int32(int(n)), fptr(out)))
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
        check(self._lib.rlc_kl_update(self._h, ctypes.c_int32(int(n_updates)), iptr(idx) if idx is not None else None,
                                      fptr(e) if e is not None else None))

    def update_batch(self, agent, states, actions, next_states, rewards, gammas, eps=None):
        r = f64(rewards).reshape(-1)
        n = r.size
        s, s2 = f64(states).reshape(n, self.S), f64(next_states).reshape(n, self.S)
        a, g = f64(actions).reshape(n, self.A), f64(gammas).reshape(n)
        e = None if eps is None else np.ascontiguousarray(eps, np.float32).reshape(n, self.A)
        check(self._lib.rlc_kl_update_batch(self._h, int(agent), ctypes.c_int32(n), dptr(s), dptr(a), dptr(s2), dptr(r),
                                            dptr(g), fptr(e) if e is not None else None))

    KERNEL = {"auto": 0, "generic": 1, "mfma": 2}

    def set_kernel(self, name):
        check(self._lib.rlc_kl_set_kernel(self._h, self.KERNEL[name]))

    def kernel_in_use(self):
        out = ctypes.c_int32(0)
        check(self._lib.rlc_kl_get_kernel(self._h, ctypes.byref(out)))
        return {v: k for k, v in self.KERNEL.items()}[out.value]

    def enable_grad_taps(self, on=True):
        check(self._lib.rlc_kl_enable_grad_taps(self._h, 1 if on else 0))

    def last_tap(self, agent, which):
        n = {"q": self.B, "v": self.B, "logp": self.B, "q_pi": self.B, "loss": 3, "grads": self.P,
             "intgrl_q": self.B * self.n_nodes}[which]
        out = np.empty(n, np.float32)
        check(self._lib.rlc_kl_last_tap(self._h, int(agent), self.TAP[which], fptr(out), ctypes.c_int64(n)))
        return out
