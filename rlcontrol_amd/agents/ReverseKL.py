"""ReverseKL agent on MI355X (mirrors agents/ReverseKL.py:13-97 + agents/network/reversekl_network.py).

Same construction (``ReverseKL(config)`` from the Config main.py builds out of jsonfiles/agent/reverse_kl.json) and the
same ``start/step/update/reset`` behaviour: training actions are samples ``tanh(mean + std*eps) * action_max[0]``
(above one action dimension the reference's MultivariateNormal(mean, diag_embed(std)) makes that ``sqrt(std)*eps``)
(external exploration raises NotImplementedError as in the reference, :38-39), evaluation uses ``tanh(mean)`` unless
``sample_for_eval == "True"``; every update is update_network followed by update_target_network (:83-93).
The eps stream is ``numpy.RandomState(random_seed)`` on the host (the reference draws it from torch's global
generator: statistical parity only).  ``use_true_q`` raises NotImplementedError as in the reference (:85-86);
``write_plot`` (getQFunction / getPolicyFunction for utils.plot_utils) is not provided.
"""
import numpy as np

from .base_agent import BaseAgent
from .network.base_network_manager import BaseNetwork_Manager
from ..hip_kl import KLPopulation, init_params


class KL_Network_Manager(BaseNetwork_Manager):
    """What ReverseKL_Network_Manager and ForwardKL_Network_Manager share (the two reference files differ in the
    network class they construct and in one plot title: ``diff agents/ReverseKL.py agents/ForwardKL.py``)."""
    KIND = None

    def __init__(self, config):
        super(KL_Network_Manager, self).__init__(config)
        self.rng = np.random.RandomState(config.random_seed)
        self.sample_for_eval = config.sample_for_eval == "True"
        self.use_true_q = config.use_true_q == "True"
        if self.KIND == "forward" and config.optim_type != "intg":
            # forwardkl_network.py:153-158: 'll' raises, any other name leaves policy_loss undefined
            raise NotImplementedError("ForwardKL implements optim_type 'intg' only")
        self.population = KLPopulation(
            self.KIND, n_agents=1, state_dim=config.state_dim, action_dim=config.action_dim,
            actor_l1_dim=config.actor_l1_dim, actor_l2_dim=config.actor_l2_dim,
            critic_l1_dim=config.critic_l1_dim, critic_l2_dim=config.critic_l2_dim,
            batch_size=config.batch_size, buffer_size=int(config.buffer_size), tau=config.tau,
            action_max0=float(np.asarray(config.action_max).reshape(-1)[0]),
            pi_lr=config.pi_lr, qf_vf_lr=config.qf_vf_lr, entropy_scale=config.entropy_scale,
            seeds=[np.uint64(config.random_seed)], n_param=config.N_param, optim_type=config.optim_type,
            q_update_type=config.q_update_type, device=int(getattr(config, "device", 0)),
            l_param=getattr(config, "l_param", None), action_max=config.action_max)
        theta0 = init_params(config.state_dim, config.action_dim, config.actor_l1_dim, config.actor_l2_dim,
                             config.critic_l1_dim, config.critic_l2_dim, config.random_seed)
        self.population.set_params(0, theta0, init_target=True)
        # optional json key "hip_split": latency mode, this one agent's action integral over that many CUs (the GPU must
        # not be shared while it learns; rlc_kl_set_split)
        split = int(getattr(config, "hip_split", 1))
        if split > 1:
            self.population.set_split(split)

    def device_replay(self):
        return (self.population, 0)

    def _eps(self, n):
        return self.rng.standard_normal((n, self.action_dim)).astype(np.float32)

    # ---- the training sample for the next state, queued behind the update (see agents/SoftActorCritic.py) ----
    queues_next_action = True
    _queued_state = None

    def _queue_sample(self, next_state):
        self._queued_rng = self.rng.get_state()
        self.population.act_queue(np.expand_dims(next_state, 0), sample=True, eps=self._eps(1))
        self._queued_state = np.array(next_state, np.float64)

    def _drop_queued(self):
        if self._queued_state is not None:
            self.rng.set_state(self._queued_rng)
            self._queued_state = None

    def _train_sample(self, state):
        queued, self._queued_state = self._queued_state, None
        if queued is not None:
            if np.array_equal(queued, np.asarray(state, np.float64)):
                return self.population.act_fetch(1)[0]
            self.rng.set_state(self._queued_rng)
        return self.population.act(np.expand_dims(state, 0), sample=True, eps=self._eps(1))[0]

    def take_action(self, state, is_train, is_start):
        if is_train:
            if is_start:
                self.train_ep_count += 1
            self.train_global_steps += 1
            if self.use_external_exploration:
                raise NotImplementedError
            return self._train_sample(state)
        self._drop_queued()
        if self.sample_for_eval:
            chosen = self.population.act(np.expand_dims(state, 0), sample=True, eps=self._eps(1))[0]
        else:
            chosen = self.population.act(np.expand_dims(state, 0), sample=False)[0]
        if is_start:
            self.eval_ep_count += 1
        self.eval_global_steps += 1
        return chosen

    def update_network(self, state_batch, action_batch, next_state_batch, reward_batch, gamma_batch):
        if self.use_true_q:
            raise NotImplementedError
        self._drop_queued()
        n = len(np.reshape(reward_batch, -1))
        self.population.update_batch(0, state_batch, action_batch, next_state_batch, reward_batch, gamma_batch,
                                     eps=self._eps(n))

    def update_from_replay(self, logical_indices, next_state=None):
        if self.use_true_q:
            raise NotImplementedError
        self._drop_queued()
        self.population.update(1, host_indices=logical_indices, eps=self._eps(len(logical_indices)))
        if next_state is not None:
            self._queue_sample(next_state)


class ReverseKL_Network_Manager(KL_Network_Manager):
    KIND = "reverse"


class ReverseKL(BaseAgent):
    def __init__(self, config):
        network_manager = ReverseKL_Network_Manager(config)
        super(ReverseKL, self).__init__(config, network_manager)
