"""Experiment.run() of the reference (experiment.py:52-217) for a whole population, environment on the GPU.

``DeviceExperiment(population, env_params, gamma, warmup_steps)`` drives the C ABI's rollout block
(include/rlcontrol_hip.h): every agent of a ``DDPGPopulation`` (act + OU noise), ``SACPopulation``
(reparameterised sample of the policy) or ``NAFPopulation`` (draw from N(mu, noise_scale (L L^T)^-1)) runs its
own train / evaluate loop on the device -- act, Pendulum step,
replay insert, gated fused update, periodic greedy evaluation -- with no host round trip per step.  ``run()`` returns, per agent, the reference's 9-tuple
(train_rewards_per_episode, eval_rewards_per_episode, train_steps_per_episode, eval_steps_per_episode,
timesteps_at_eval, cum_train_time, cum_eval_time, train_episodes, train_cum_steps).

Differences from the host-driven ``Experiment`` (documented, not hidden): the random streams are the device's
Philox streams (environment resets, OU normals, minibatch indices) instead of numpy's MT19937 ones, and the
two wall-clock entries of the tuple are the population's total split by the fraction of steps, since
training and evaluation of all agents are interleaved on one stream.
"""
import ctypes
import time

import numpy as np

from . import _lib
from ._lib import check, dptr, iptr


class DeviceExperiment(object):
    def __init__(self, population, env_params, gamma=0.99, warmup_steps=0, max_train_episodes=None, noise_scale=None):
        name = env_params['environment']
        if name not in _lib.ENV_IDS:
            raise RuntimeError("environment %r is not simulated on the device (built in: %s)"
                               % (name, ", ".join(sorted(_lib.ENV_IDS))))
        self.pop = population
        self.name = name
        self.total_steps_limit = int(env_params['TotalMilSteps'] * 1000000)
        self.eval_interval = int(env_params['EvalIntervalMilSteps'] * 1000000)
        self.eval_episodes = int(env_params['EvalEpisodes'])
        self.episode_steps_limit = 200 if env_params['EpisodeSteps'] == -1 else int(env_params['EpisodeSteps'])
        if max_train_episodes is None:
            max_train_episodes = max(1, min(self.total_steps_limit, 1 << 20))
        cfg = _lib.rlc_rollout_config()
        cfg.env_id = _lib.ENV_IDS[name]
        cfg.episode_steps_limit = self.episode_steps_limit
        cfg.total_steps_limit = self.total_steps_limit
        cfg.eval_interval = self.eval_interval
        cfg.eval_episodes = self.eval_episodes
        cfg.warmup_steps = int(warmup_steps)
        cfg.max_train_episodes = int(max_train_episodes)
        cfg.gamma = float(gamma)
        self._max_ep = int(max_train_episodes)
        kind = type(population).__name__
        self._prefix = {"SACPopulation": "rlc_sac", "NAFPopulation": "rlc_naf", "KLPopulation": "rlc_kl"}.get(kind, "rlc_ddpg")
        create = getattr(population._lib, self._prefix + "_rollout_create")
        if self._prefix == "rlc_naf":
            if noise_scale is None:
                raise ValueError("a NAF population needs noise_scale (per agent or one value)")
            self._ns = np.ascontiguousarray(np.broadcast_to(np.asarray(noise_scale, np.float32).reshape(-1),
                                                            (population.n_agents,)))
            check(create(population._h, ctypes.byref(cfg), self._ns.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
        else:
            check(create(population._h, ctypes.byref(cfg)))
        self.total_steps = 0
        self.wall = 0.0

    def advance(self, n_steps):
        """up to n_steps more training steps for every agent; returns the steps taken so far"""
        out = ctypes.c_int64(0)
        t0 = time.time()
        check(getattr(self.pop._lib, self._prefix + "_rollout_run")(self.pop._h, ctypes.c_int64(int(n_steps)),
                                                                    ctypes.byref(out)))
        self.wall += time.time() - t0
        self.total_steps = int(out.value)
        return self.total_steps

    def counts(self, agent):
        ne, nv, ts = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        check(self.pop._lib.rlc_rollout_counts(self.pop._h, int(agent), ctypes.byref(ne), ctypes.byref(nv),
                                               ctypes.byref(ts)))
        return int(ne.value), int(nv.value), int(ts.value)

    def train_log(self, agent):
        ne = min(self.counts(agent)[0], self._max_ep)
        ret, ln, cum = np.empty(ne), np.empty(ne, np.int32), np.empty(ne, np.int64)
        check(self.pop._lib.rlc_rollout_train_log(self.pop._h, int(agent), ctypes.c_int64(ne), dptr(ret),
                                                  ln.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), iptr(cum)))
        return ret, ln, cum

    def eval_log(self, agent):
        nv = self.counts(agent)[1]
        ret = np.empty((nv, self.eval_episodes))
        ln = np.empty((nv, self.eval_episodes), np.int32)
        check(self.pop._lib.rlc_rollout_eval_log(self.pop._h, int(agent), ctypes.c_int64(nv), dptr(ret),
                                                 ln.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
        return ret, ln

    def observation(self, agent):
        obs = np.empty(self.pop.S)
        step = ctypes.c_int32(0)
        check(self.pop._lib.rlc_rollout_observation(self.pop._h, int(agent), dptr(obs), ctypes.byref(step)))
        return obs, int(step.value)

    def run(self, chunk=5000, progress=None):
        while self.total_steps < self.total_steps_limit:
            self.advance(min(chunk, self.total_steps_limit - self.total_steps))
            if progress is not None:
                progress(self.total_steps)
        return self.results()

    def results(self):
        out = []
        eval_steps_total = None
        for a in range(self.pop.n_agents):
            tr, tl, tc = self.train_log(a)
            er, el = self.eval_log(a)
            n_ep_started = len(tr) + (1 if (len(tc) == 0 or tc[-1] < self.total_steps) and self.total_steps > 0 else 0)
            if eval_steps_total is None:
                eval_steps_total = float(el.sum())
            frac_eval = eval_steps_total / max(eval_steps_total + self.total_steps, 1.0)
            timesteps_at_eval = [i * self.eval_interval for i in range(er.shape[0])]
            out.append((list(tr), [list(r) for r in er], [int(x) for x in tl], [[int(x) for x in r] for r in el],
                        timesteps_at_eval, self.wall * (1.0 - frac_eval), self.wall * frac_eval, n_ep_started,
                        [int(x) for x in tc]))
        return out
