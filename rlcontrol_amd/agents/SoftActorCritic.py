"""SoftActorCritic agent on MI355X (mirrors agents/SoftActorCritic.py:15-137 + agents/network/sac_network.py).

Same construction (``SoftActorCritic(config)`` from the Config main.py builds out of jsonfiles/agent/sac.json)
and the same ``start/step/update/reset`` behaviour: training actions are reparameterised samples
``tanh(mu + eps*std) * action_max[0]`` (exploration_policy 'none'), evaluation uses the mean action unless
``sample_for_eval == "True"``.  The eps stream is ``numpy.RandomState(random_seed)`` on the host (the reference
draws it with tf.random_normal, which cannot be reproduced without TensorFlow: statistical parity only).
``use_true_q`` (restoring the Bimodal critic checkpoints) is out of scope.
"""
import numpy as np

from .base_agent import BaseAgent
from .network.base_network_manager import BaseNetwork_Manager, check_norm_type
from ..hip_sac import SACPopulation, init_params


class SoftActorCritic_Network_Manager(BaseNetwork_Manager):
    queues_next_action = True        # update_from_replay(indices, next_state=...) queues the forward step() will fetch

    def __init__(self, config):
        super(SoftActorCritic_Network_Manager, self).__init__(config)
        self.rng = np.random.RandomState(config.random_seed)
        self.sample_for_eval = config.sample_for_eval == "True"
        if getattr(config, "use_true_q", "False") == "True":
            raise NotImplementedError("use_true_q (Bimodal checkpoints) is outside the accelerated path")
        if config.norm_type == 'none':
            # the reference leaves `inputs` undefined in that case (quirk Q10, sac_network.py:175-178)
            raise ValueError("SoftActorCritic needs norm_type != 'none'")
        check_norm_type(config, "SoftActorCritic", ('input_norm', 'layer'))
        self.population = SACPopulation(
            n_agents=1, state_dim=config.state_dim, action_dim=config.action_dim,
            actor_l1_dim=config.actor_l1_dim, actor_l2_dim=config.actor_l2_dim,
            critic_l1_dim=config.critic_l1_dim, critic_l2_dim=config.critic_l2_dim,
            batch_size=config.batch_size, buffer_size=int(config.buffer_size), tau=config.tau,
            state_min0=float(np.asarray(config.state_min).reshape(-1)[0]),
            state_max0=float(np.asarray(config.state_max).reshape(-1)[0]),
            action_max0=float(np.asarray(config.action_max).reshape(-1)[0]),
            pi_lr=config.pi_lr, qf_vf_lr=config.qf_vf_lr, entropy_scale=config.entropy_scale,
            seeds=[np.uint64(config.random_seed)], clip_state=True, device=int(getattr(config, "device", 0)),
            norm_type=config.norm_type)
        theta0 = init_params(config.state_dim, config.action_dim, config.actor_l1_dim, config.actor_l2_dim,
                             config.critic_l1_dim, config.critic_l2_dim, config.random_seed, config.norm_type)
        self.population.set_params(0, theta0, init_target=True)

    def device_replay(self):
        return (self.population, 0)

    def _eps(self, n):
        return self.rng.standard_normal((n, self.action_dim)).astype(np.float32)

    # ---- the training sample for the next state, queued behind the update (one synchronisation per environment step) ----
    _queued_state = None

    def _queue_sample(self, next_state):
        """Experiment asks step(next_state, is_train=True) next (experiment.py:132-135): that forward is queued now, with
        the eps take_action would draw -- it is the next draw of the stream either way.  If something else is asked
        first (an evaluation episode, another state), _drop_queued puts the stream back and the forward is discarded."""
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
                self._drop_queued()
                greedy = self.population.act(np.expand_dims(state, 0), sample=False)[0]
                return self.exploration_policy.generate(greedy, self.train_global_steps)
            return self._train_sample(state)
        self._drop_queued()
        if is_start:
            self.eval_ep_count += 1
        self.eval_global_steps += 1
        if self.sample_for_eval:
            return self.population.act(np.expand_dims(state, 0), sample=True, eps=self._eps(1))[0]
        return self.population.act(np.expand_dims(state, 0), sample=False)[0]

    def update_network(self, state_batch, action_batch, next_state_batch, reward_batch, gamma_batch):
        self._drop_queued()
        n = len(np.reshape(reward_batch, -1))
        self.population.update_batch(0, state_batch, action_batch, next_state_batch, reward_batch, gamma_batch,
                                     eps=self._eps(n))

    def update_from_replay(self, logical_indices, next_state=None):
        self._drop_queued()
        self.population.update(1, host_indices=logical_indices, eps=self._eps(len(logical_indices)))
        if next_state is not None and not self.use_external_exploration:
            self._queue_sample(next_state)


class SoftActorCritic(BaseAgent):
    def __init__(self, config):
        network_manager = SoftActorCritic_Network_Manager(config)
        super(SoftActorCritic, self).__init__(config, network_manager)
