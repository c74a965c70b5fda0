"""Common part of every population handle (rlc_handle): lifetime, replay ring, timing.

One handle = a population of independent agents of ONE algorithm on one MI355X (include/rlcontrol_hip.h).
The per-algorithm classes (hip_ddpg.DDPGPopulation, hip_sac.SACPopulation) add networks and learning."""
import ctypes

import numpy as np

from . import _lib
from ._lib import check, dptr, f64, iptr


class Population(object):
    def _init_base(self, n_agents, state_dim, action_dim, batch_size):
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        self.n_agents = int(n_agents)
        self.S, self.A, self.B = int(state_dim), int(action_dim), int(batch_size)

    # ---- lifetime -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.rlc_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self._lib.rlc_sync(self._h))

    # ---- replay ---------------------------------------------------------------------------
    def replay_add(self, agent, state, action, reward, next_state, transition_gamma):
        s, a, s2 = f64(state).reshape(-1), f64(action).reshape(-1), f64(next_state).reshape(-1)
        if s.size != self.S or s2.size != self.S or a.size != self.A:
            raise ValueError("transition shapes do not match state_dim/action_dim")
        check(self._lib.rlc_replay_add(self._h, int(agent), dptr(s), dptr(a), ctypes.c_double(float(reward)),
                                       dptr(s2), ctypes.c_double(float(transition_gamma))))

    def replay_add_batch(self, agent, states, actions, rewards, next_states, gammas):
        r = f64(rewards).reshape(-1)
        n = r.size
        s, s2 = f64(states).reshape(n, self.S), f64(next_states).reshape(n, self.S)
        a, g = f64(actions).reshape(n, self.A), f64(gammas).reshape(n)
        check(self._lib.rlc_replay_add_batch(self._h, int(agent), ctypes.c_int64(n), dptr(s), dptr(a), dptr(r),
                                             dptr(s2), dptr(g)))

    def replay_fill_all_dev(self, n, s_ptr, a_ptr, r_ptr, s2_ptr, g_ptr):
        """device pointers (ints), e.g. torch tensor .data_ptr(): fp32 s/a/s2, fp64 r/gamma"""
        vp = ctypes.c_void_p
        check(self._lib.rlc_replay_fill_all_dev(self._h, ctypes.c_int64(int(n)), vp(s_ptr), vp(a_ptr), vp(r_ptr),
                                                vp(s2_ptr), vp(g_ptr)))

    def replay_size(self, agent):
        out = ctypes.c_int64(0)
        check(self._lib.rlc_replay_size(self._h, int(agent), ctypes.byref(out)))
        return int(out.value)

    def replay_gather(self, agent, logical_idx):
        idx = np.ascontiguousarray(logical_idx, np.int64).reshape(-1)
        k = idx.size
        s, s2 = np.empty((k, self.S)), np.empty((k, self.S))
        a, r, g = np.empty((k, self.A)), np.empty(k), np.empty(k)
        check(self._lib.rlc_replay_gather(self._h, int(agent), iptr(idx), ctypes.c_int32(k), dptr(s), dptr(a),
                                          dptr(r), dptr(s2), dptr(g)))
        return s, a, r, s2, g

    def replay_sample_indices(self, agent, k):
        out = np.empty(int(k), np.int64)
        check(self._lib.rlc_replay_sample_indices(self._h, int(agent), ctypes.c_int32(int(k)), iptr(out)))
        return out

    # ---- timing ---------------------------------------------------------------------------
    def timer_begin(self):
        check(self._lib.rlc_timer_begin(self._h))

    def timer_end(self):
        ms = ctypes.c_float(0.0)
        check(self._lib.rlc_timer_end(self._h, ctypes.byref(ms)))
        return float(ms.value)
