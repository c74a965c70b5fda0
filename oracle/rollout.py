"""TEST INFRASTRUCTURE (oracle) -- never imported by the product path.

CPU restatement of the on-device experiment loop (rlcontrol_amd/csrc/rollout_kernels.hip +
rlc_api_rollout.hip), i.e. of the reference's Experiment.run / run_episode_train / eval
(experiment.py:52-217), BaseAgent.update / learn (agents/base_agent.py:54-70), the OU process
(utils/exploration_policy.py:18-24) and the DDPG manager (agents/DDPG.py:34-95), with the random draws
taken from the Philox streams the device uses (oracle/philox.py) so that device and CPU runs are comparable
step by step.  Written sequentially, in the reference's own order (act AFTER the update, eval inside the
training episode, OU reset by eval: quirk Q8) -- the device fuses "act after update t" into step t+1 and the
test checks the two orderings give the same trajectory.

Pendulum-v0 is restated from the public gym 0.18.0 definition (third-party, absent: parity unpinned), float64.
"""
import math

import numpy as np

from . import philox
from .ddpg import DDPGOracle

KEY_SAC_EPS = 0x9E3779B97F4A7C15     # sac_policy.h: the agent's N(0,1) stream (acting and minibatch draws)
KEY_NAF_EPS = 0x4E41465F4E4F4953     # naf_policy.h: the N(0,1) stream of NAF's exploration draw


class Pendulum(object):
    def __init__(self, key):
        self.key = key
        self.resets = 0
        self.th = 0.0
        self.thdot = 0.0

    def _obs(self):
        return np.array([math.cos(self.th), math.sin(self.th), self.thdot])

    def reset_at(self, ctr):
        p = philox.philox4x32_10(self.key, ctr, 0)
        self.th = -math.pi + 2.0 * math.pi * philox.uniform01_double(p[0], p[1])
        self.thdot = -1.0 + 2.0 * philox.uniform01_double(p[2], p[3])
        return self._obs()

    def reset(self):
        o = self.reset_at(self.resets)
        self.resets += 1
        return o

    def step(self, action):
        th, thdot = self.th, self.thdot
        u = min(max(float(action[0]), -2.0), 2.0)
        wrapped = ((th + math.pi) % (2.0 * math.pi)) - math.pi
        cost = wrapped * wrapped + 0.1 * thdot * thdot + 0.001 * (u * u)
        nthdot = thdot + (-3.0 * 10.0 / (2.0 * 1.0) * math.sin(th + math.pi) + 3.0 / (1.0 * 1.0 * 1.0) * u) * 0.05
        nth = th + nthdot * 0.05
        nthdot = min(max(nthdot, -8.0), 8.0)
        self.th, self.thdot = nth, nthdot
        return self._obs(), -cost


class RolloutOracle(object):
    """One agent of the on-device loop."""

    def __init__(self, dims, theta, actor_lr, critic_lr, tau, state_min, state_max, action_min, action_max, seed,
                 batch_size, buffer_size, gamma, warmup_steps, episode_limit, total_steps, eval_interval,
                 eval_episodes, ou_theta=0.15, ou_mu=0.0, ou_sigma=0.2, clip_state=True):
        self.net = self._make_net(dims, theta, actor_lr, critic_lr, tau, state_min, state_max, action_max, clip_state)
        self.seed = int(seed)
        self.B, self.cap = int(batch_size), int(buffer_size)
        self.gamma, self.warmup = float(gamma), int(warmup_steps)
        self.limit, self.total_limit = int(episode_limit), int(total_steps)
        self.eval_interval, self.eval_episodes = int(eval_interval), int(eval_episodes)
        self.amin = np.asarray(action_min, np.float32).reshape(-1)
        self.amax = np.asarray(action_max, np.float32).reshape(-1)
        f = np.float32
        self.ou_theta, self.ou_mu, self.ou_sigma = f(ou_theta), f(ou_mu), f(ou_sigma)
        self.A = dims.A if hasattr(dims, "A") else dims.t[1]      # Dims has .A; SacDims / NafDims keep a tuple .t
        self.noise = np.full(self.A, self.ou_mu, np.float32)
        self.noise_ctr = 0
        self.sample_ctr = 0
        self.replay = []                 # oldest first: (s32, a32, r64, s2_32, g64)
        self.train_env = Pendulum(self.seed ^ philox.KEY_ENV_TRAIN)
        self.test_env = Pendulum(self.seed ^ philox.KEY_ENV_TEST)
        self.total = 0
        self.evals = 0
        self.train_ret, self.train_len, self.train_cum = [], [], []
        self.eval_ret, self.eval_len, self.timesteps_at_eval = [], [], []
        self.n_updates = 0

    def _make_net(self, dims, theta, actor_lr, critic_lr, tau, state_min, state_max, action_max, clip_state):
        return DDPGOracle(dims, theta, actor_lr, critic_lr, tau, state_min, state_max, action_max, clip_state)

    def _learn(self, rows):
        self.net.update(np.array([r[0] for r in rows]), np.array([r[1] for r in rows]),
                        np.array([r[3] for r in rows]), np.array([r[2] for r in rows]),
                        np.array([r[4] for r in rows]))

    # -- agent ---------------------------------------------------------------------------------
    def agent_reset(self):
        self.noise[:] = self.ou_mu

    def act(self, obs, is_train):
        greedy = self.net.act(np.asarray(obs, np.float64).astype(np.float32).reshape(1, -1))[0]
        if not is_train:
            return greedy
        out = np.empty(self.A, np.float32)
        for j in range(self.A):
            p = philox.philox4x32_10(self.seed ^ philox.KEY_OU, self.noise_ctr, j // 2)
            z = philox.normal2(p)[j & 1]
            n = self.noise[j]
            n = np.float32(n + np.float32(np.float32(self.ou_mu + np.float32(self.ou_sigma * z)) - np.float32(n * self.ou_theta)))
            self.noise[j] = n
            out[j] = min(max(np.float32(greedy[j] + n), self.amin[j]), self.amax[j])
        self.noise_ctr += 1
        return out

    def update(self, obs, obs_n, reward, action, done, truncated):
        if not truncated:
            g = 0.0 if done else self.gamma
            self.replay.append((np.asarray(obs, np.float64).astype(np.float32), np.asarray(action, np.float32).copy(),
                                float(reward), np.asarray(obs_n, np.float64).astype(np.float32), g))
            if len(self.replay) > self.cap:
                self.replay.pop(0)
        if len(self.replay) > max(self.warmup, self.B):
            idx = philox.sample_distinct(len(self.replay), self.B, self.seed, self.sample_ctr)
            self.sample_ctr += 1
            rows = [self.replay[i] for i in idx]
            self._learn(rows)
            self.n_updates += 1

    # -- experiment ------------------------------------------------------------------------------
    def eval(self):
        rets, lens = [], []
        for e in range(self.eval_episodes):
            obs = self.test_env.reset_at(self.evals * self.eval_episodes + e)
            self.agent_reset()
            ret, steps, done = 0.0, 0, False
            action = self.act(obs, False)
            while not (done or steps == self.limit):
                obs, r = self.test_env.step(action)
                steps += 1
                done = steps >= self.limit
                ret += r
                if not done:
                    action = self.act(obs, False)
            rets.append(ret)
            lens.append(steps)
        self.eval_ret.append(rets)
        self.eval_len.append(lens)
        self.evals += 1

    def run(self, max_steps=None):
        stop = self.total_limit if max_steps is None else min(self.total_limit, max_steps)
        self.eval()
        self.timesteps_at_eval.append(self.total)
        while self.total < stop:
            obs = self.train_env.reset()
            self.agent_reset()
            ret, done, step = 0.0, False, 0
            action = self.act(obs, True)
            while not (done or step == self.limit or self.total == stop):
                step += 1
                self.total += 1
                obs_n, r = self.train_env.step(action)
                done = step >= self.limit
                ret += r
                truncated = bool(done and step == self.limit)
                self.update(obs, obs_n, r, action, done, truncated)
                if not done:
                    action = self.act(obs_n, True)
                obs = obs_n
                if self.total % self.eval_interval == 0:
                    self.timesteps_at_eval.append(self.total)
                    self.eval()
            if done or step == self.limit:
                self.train_ret.append(ret)
                self.train_len.append(step)
                self.train_cum.append(self.total)
        self.last_obs, self.last_step = obs, step
        return self


class VariantRolloutOracle(RolloutOracle):
    """DDPG with norm_type 'layer' and / or separate actor / critic networks in the on-device loop: the same loop around
    oracle/ddpg_variants.py's network (dims is a VDims)"""

    def _make_net(self, dims, theta, actor_lr, critic_lr, tau, state_min, state_max, action_max, clip_state):
        from .ddpg_variants import DDPGVariantOracle
        return DDPGVariantOracle(dims, np.asarray(theta, np.float32), actor_lr, critic_lr, tau, state_min, state_max,
                                 action_max, clip_state)


class SacRolloutOracle(RolloutOracle):
    """One SoftActorCritic agent of the on-device loop (sac_rollout_device.h): training actions are reparameterised
    samples of the current policy, evaluation uses the mean action, no exploration-noise state."""

    def __init__(self, dims, theta, pi_lr, qv_lr, alpha, tau, smin0, smax0, amax0, seed, batch_size, buffer_size, gamma,
                 warmup_steps, episode_limit, total_steps, eval_interval, eval_episodes):
        self._sac = (pi_lr, qv_lr, alpha, smin0, smax0, amax0)
        RolloutOracle.__init__(self, dims, theta, 0.0, 0.0, tau, None, None, [-amax0], [amax0], seed, batch_size,
                               buffer_size, gamma, warmup_steps, episode_limit, total_steps, eval_interval, eval_episodes)

    def _make_net(self, dims, theta, actor_lr, critic_lr, tau, state_min, state_max, action_max, clip_state):
        from .sac import SACOracle
        pi_lr, qv_lr, alpha, smin0, smax0, amax0 = self._sac
        return SACOracle(dims, theta, pi_lr, qv_lr, alpha, tau, smin0, smax0, amax0)

    def agent_reset(self):
        pass

    def _normal(self, ctr_hi_base, k):
        p = philox.philox4x32_10(self.seed ^ KEY_SAC_EPS, self.noise_ctr, ctr_hi_base + (k >> 1))
        return philox.normal2(p)[k & 1]

    def act(self, obs, is_train):
        x = np.asarray(obs, np.float64).astype(np.float32).reshape(1, -1)
        if not is_train:
            return self.net.act(x)[0]
        eps = np.array([[self._normal(0x4000000000000000, j) for j in range(self.A)]], np.float32)
        self.noise_ctr += 1
        return self.net.act(x, eps=eps)[0]

    def _learn(self, rows):
        B = len(rows)
        eps = np.array([[self._normal(0, b * self.A + j) for j in range(self.A)] for b in range(B)], np.float32)
        self.noise_ctr += 1
        self.net.update(np.array([r[0] for r in rows]), np.array([r[1] for r in rows]),
                        np.array([r[3] for r in rows]), np.array([r[2] for r in rows]),
                        np.array([r[4] for r in rows]), eps)


class KlRolloutOracle(SacRolloutOracle):
    """One ReverseKL / ForwardKL agent of the on-device loop: the step is SoftActorCritic's (sampled training action,
    mean evaluation action, the same two Philox sub-streams for acting and minibatch draws) on the torch restatement
    of the KL networks (oracle/kl_torch.py); the states enter unclipped."""

    def __init__(self, kind, dims, theta, pi_lr, qv_lr, alpha, tau, amax0, n_param, seed, batch_size, buffer_size, gamma,
                 warmup_steps, episode_limit, total_steps, eval_interval, eval_episodes, optim_type="intg",
                 q_update_type="non_sac"):
        self._kl = (kind, pi_lr, qv_lr, alpha, amax0, n_param, optim_type, q_update_type)
        RolloutOracle.__init__(self, dims, theta, 0.0, 0.0, tau, None, None, [-amax0], [amax0], seed, batch_size,
                               buffer_size, gamma, warmup_steps, episode_limit, total_steps, eval_interval, eval_episodes)

    def _make_net(self, dims, theta, actor_lr, critic_lr, tau, state_min, state_max, action_max, clip_state):
        from .kl_torch import KLOracle
        kind, pi_lr, qv_lr, alpha, amax0, n_param, optim_type, q_update_type = self._kl
        return KLOracle(kind, dims, theta, pi_lr, qv_lr, alpha, tau, amax0, n_param, optim_type, q_update_type)


class NafRolloutOracle(RolloutOracle):
    """One NAF agent of the on-device loop (naf_rollout_device.h): the training action is
    mu + sqrt(noise_scale) * L^-T z (a draw from N(mu, noise_scale (L L^T)^-1), naf_network.py:152-176) clipped to
    the action bounds, evaluation uses the greedy action."""

    def __init__(self, dims, theta, lr, tau, state_min, state_max, action_max, noise_scale, seed, batch_size,
                 buffer_size, gamma, warmup_steps, episode_limit, total_steps, eval_interval, eval_episodes,
                 norm_type="input_norm", action_min=None):
        self._naf = (lr, norm_type)
        self.noise_scale = np.float32(noise_scale)
        amax = np.asarray(action_max, np.float32).reshape(-1)
        amin = -amax if action_min is None else np.asarray(action_min, np.float32).reshape(-1)
        RolloutOracle.__init__(self, dims, theta, 0.0, 0.0, tau, state_min, state_max, amin, amax, seed, batch_size,
                               buffer_size, gamma, warmup_steps, episode_limit, total_steps, eval_interval, eval_episodes)

    def _make_net(self, dims, theta, actor_lr, critic_lr, tau, state_min, state_max, action_max, clip_state):
        if self._naf[1] == "layer":      # torch restatement (oracle/naf_variants.py); .theta is a tensor there
            from .naf_variants import NafVariantOracle
            return NafVariantOracle(dims.t, theta, self._naf[0], tau, state_min, state_max, action_max, "layer", clip_state)
        from .naf import NAFOracle
        return NAFOracle(dims, theta, self._naf[0], tau, state_min, state_max, action_max, clip_state)

    def agent_reset(self):
        pass

    def act(self, obs, is_train):
        x = np.asarray(obs, np.float64).astype(np.float32).reshape(1, -1)
        mu, lcols = self.net.act(x)
        mu, lcols = mu[0], lcols[0]
        if not is_train:
            return mu
        A, f = self.A, np.float32
        Lm = np.zeros((A, A), np.float32)
        p = 0
        for c in range(A):
            for i in range(c, A):
                Lm[i, c] = lcols[p]
                p += 1
        y = np.zeros(A, np.float32)
        sc = f(np.sqrt(self.noise_scale, dtype=np.float32))
        for j in range(A):
            z = philox.normal2(philox.philox4x32_10(self.seed ^ KEY_NAF_EPS, self.noise_ctr, j >> 1))[j & 1]
            y[j] = f(sc * z)
        for i in range(A - 1, -1, -1):
            s = y[i]
            for j in range(i + 1, A):
                s = f(s - f(Lm[j, i] * y[j]))
            y[i] = f(s / Lm[i, i])
        self.noise_ctr += 1
        return np.minimum(np.maximum(mu + y, self.amin), self.amax).astype(np.float32)
