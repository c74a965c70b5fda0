"""DDPG agent on MI355X (mirrors agents/DDPG.py:16-103 + agents/network/hydra_ddpg_network.py).

``DDPG(config)`` is constructed from the same ``Config`` the reference builds in main.py:160-163
(env facts + json sweep values + CLI flags) and answers the same ``start/step/update/reset`` calls.
The "hydra" network (shared first layer, actor head, critic head with the action concatenated
last), its target copy, both TF-style Adam optimizers and the replay ring all live inside one
``rlc_ddpg`` handle (include/rlcontrol_hip.h); this file only moves numpy arrays across the ABI.

Manager-level API kept: ``take_action(state, is_train, is_start)``,
``update_network(state, action, next_state, reward, gamma)``, ``reset()``, ``input_norm``.
Added for the fused path: ``update_from_replay(logical_indices)`` and ``device_replay()``.
"""
import numpy as np

from .base_agent import BaseAgent
from .network.base_network_manager import BaseNetwork_Manager, check_norm_type
from ..hip_ddpg import DDPGPopulation, init_params


class DDPG_Network_Manager(BaseNetwork_Manager):
    queues_next_action = True        # update_from_replay(indices, next_state=...) queues the forward step() will fetch

    def __init__(self, config):
        super(DDPG_Network_Manager, self).__init__(config)
        check_norm_type(config, "DDPG", ('none', 'input_norm', 'layer'))
        # `network: separate` selects the reference's actor_network.py / critic_network.py pair (commented out in
        # agents/DDPG.py:8-9,24-25); the default is the hydra network it builds (agents/DDPG.py:26)
        separate = getattr(config, "network", "hydra") == "separate"
        if separate:
            # actor_network.py / critic_network.py size their first layers from actor_l1_dim / critic_l1_dim; this
            # variant (an extension: the reference ships DDPG with the hydra network only) has ONE width for both
            # trunks, shared_l1_dim -- refuse a config that asks for anything else instead of ignoring it
            for key in ("actor_l1_dim", "critic_l1_dim"):
                if hasattr(config, key) and int(getattr(config, key)) != int(config.shared_l1_dim):
                    raise ValueError("network 'separate': %s = %r differs from shared_l1_dim = %r; both first layers "
                                     "take shared_l1_dim here" % (key, getattr(config, key), config.shared_l1_dim))
        self.population = DDPGPopulation(
            n_agents=1, state_dim=config.state_dim, action_dim=config.action_dim,
            shared_l1_dim=config.shared_l1_dim, actor_l2_dim=config.actor_l2_dim,
            critic_l2_dim=config.critic_l2_dim, batch_size=config.batch_size,
            buffer_size=int(config.buffer_size), tau=config.tau,
            state_min=config.state_min, state_max=config.state_max,
            action_min=config.action_min, action_max=config.action_max,
            actor_lr=config.actor_lr, critic_lr=config.critic_lr,
            seeds=[np.uint64(config.random_seed)],
            clip_state=(config.norm_type != 'none'),
            ou_theta=config.ou_theta, ou_mu=config.ou_mu, ou_sigma=config.ou_sigma,
            device=int(getattr(config, "device", 0)),
            norm_type=config.norm_type, separate_networks=separate)
        kernel = getattr(config, "hip_kernel", "auto")
        if kernel != "auto":
            self.population.set_kernel(kernel)
        # optional json key "hip_split": latency mode, this one agent's minibatch over that many CUs (the GPU must not
        # be shared with other processes while it learns: the workgroups meet at barriers)
        split = int(getattr(config, "hip_split", 1))
        if split > 1:
            self.population.set_split(split)
        # sess.run(global_variables_initializer()) + init_target_network() (agents/DDPG.py:28-32)
        theta0 = init_params(config.state_dim, config.action_dim, config.shared_l1_dim, config.actor_l2_dim,
                             config.critic_l2_dim, config.random_seed, config.norm_type, separate)
        self.population.set_params(0, theta0, init_target=True)
        self._queued_state = None

    def device_replay(self):
        return (self.population, 0)

    def _greedy(self, state):
        # the forward for this very state may already be queued behind the last update (update_from_replay below)
        queued, self._queued_state = self._queued_state, None
        if queued is not None and np.array_equal(queued, np.asarray(state, np.float64)):
            return self.population.act_fetch(1)[0]
        return self.population.act(np.expand_dims(state, 0))[0]

    def take_action(self, state, is_train, is_start):
        greedy_action = self._greedy(state)
        if is_train:
            if is_start:
                self.train_ep_count += 1
            self.train_global_steps += 1
            if self.use_external_exploration:
                chosen_action = self.exploration_policy.generate(greedy_action, self.train_global_steps)
            else:
                chosen_action = greedy_action
        else:
            if is_start:
                self.eval_ep_count += 1
            self.eval_global_steps += 1
            chosen_action = greedy_action
        return chosen_action

    def update_network(self, state_batch, action_batch, next_state_batch, reward_batch, gamma_batch):
        self._queued_state = None
        self.population.update_batch(0, state_batch, action_batch, next_state_batch, reward_batch, gamma_batch)

    def update_from_replay(self, logical_indices, next_state=None):
        """One fused update on the device replay.  `next_state`: the observation BaseAgent.update was handed and
        Experiment will ask an action for next (experiment.py:132-135): its greedy forward is queued behind the update
        (one launch sequence, one synchronisation per environment step); exploration noise is still drawn in
        take_action, after the greedy action, as the reference does (agents/DDPG.py:36-48)."""
        self._queued_state = None
        self.population.update(1, host_indices=logical_indices)
        if next_state is not None:
            self.population.act_queue(np.expand_dims(next_state, 0))
            self._queued_state = np.array(next_state, np.float64)


class DDPG(BaseAgent):
    def __init__(self, config):
        network_manager = DDPG_Network_Manager(config)
        super(DDPG, self).__init__(config, network_manager)
