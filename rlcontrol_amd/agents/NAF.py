"""NAF agent on MI355X (mirrors agents/NAF.py:11-85 + agents/network/naf_network.py).

``NAF(config)`` is built from the Config of jsonfiles/agent/naf.json.  Training actions follow the reference's
exploration: the device returns the greedy action and the L columns, the host builds
``covmat = noise_scale * pinv(L L^T)`` and draws ``rng.multivariate_normal(greedy, covmat)`` with
``RandomState(random_seed)`` then clips to the action bounds (naf_network.py:152-176) -- same RNG stream as the
reference.  Evaluation returns the greedy action.
"""
import numpy as np

from .base_agent import BaseAgent
from .network.base_network_manager import BaseNetwork_Manager, check_norm_type
from ..hip_naf import NAFPopulation, init_params


class NAF_Network_Manager(BaseNetwork_Manager):
    queues_next_action = True        # update_from_replay(indices, next_state=...) queues the forward step() will fetch
    _queued_state = None

    def __init__(self, config):
        super(NAF_Network_Manager, self).__init__(config)
        check_norm_type(config, "NAF", ('none', 'input_norm', 'layer'))
        self.rng = np.random.RandomState(config.random_seed)      # NAF_Network.rng (naf_network.py:10)
        self.noise_scale = config.noise_scale
        self.population = NAFPopulation(
            n_agents=1, state_dim=config.state_dim, action_dim=config.action_dim, l1_dim=config.l1_dim,
            l2_dim=config.l2_dim, batch_size=config.batch_size, buffer_size=int(config.buffer_size), tau=config.tau,
            state_min=config.state_min, state_max=config.state_max, action_max=config.action_max,
            learning_rate=config.learning_rate, seeds=[np.uint64(config.random_seed)],
            clip_state=(config.norm_type != 'none'), device=int(getattr(config, "device", 0)),
            norm_type=config.norm_type)
        theta0 = init_params(config.state_dim, config.action_dim, config.l1_dim, config.l2_dim, config.random_seed,
                             config.norm_type)
        self.population.set_params(0, theta0, init_target=True)

    def device_replay(self):
        return (self.population, 0)

    def _sample_action(self, greedy, lcols):
        A = self.action_dim
        Lmat = np.zeros((A, A))
        p = 0
        for i in range(A):
            Lmat[i:, i] = lcols[p:p + A - i]
            p += A - i
        covmat = self.noise_scale * np.linalg.pinv(Lmat.dot(Lmat.T))
        sampled = self.rng.multivariate_normal(greedy.reshape(-1), covmat)
        return np.clip(sampled, self.action_min, self.action_max)

    def take_action(self, state, is_train, is_start):
        if is_train:
            if is_start:
                self.train_ep_count += 1
            self.train_global_steps += 1
            queued, self._queued_state = self._queued_state, None
            if queued is not None and np.array_equal(queued, np.asarray(state, np.float64).reshape(-1)):
                mu, lc = self.population.act_fetch(1)        # queued behind the last update (update_from_replay)
            else:
                mu, lc = self.population.act(state.reshape(-1, self.state_dim), with_lcols=True)
            if self.use_external_exploration:
                return self.exploration_policy.generate(mu, self.train_global_steps)
            return self._sample_action(mu[0].astype(np.float64), lc[0].astype(np.float64))
        self._queued_state = None
        if is_start:
            self.eval_ep_count += 1
        self.eval_global_steps += 1
        return self.population.act(state.reshape(-1, self.state_dim)).reshape(-1)

    def update_network(self, state, action, next_state, reward, gamma):
        self._queued_state = None
        self.population.update_batch(0, state, action, next_state, reward, gamma)

    def update_from_replay(self, logical_indices, next_state=None):
        """`next_state`: the observation Experiment asks an action for next (experiment.py:132-135): its greedy forward
        (mu and the L columns) is queued behind the update -- one launch sequence, one synchronisation per environment
        step; the exploration sample is still drawn on the host after the fetch, as the reference does (agents/NAF.py)."""
        self._queued_state = None
        self.population.update(1, host_indices=logical_indices)
        if next_state is not None:
            self.population.act_queue(np.asarray(next_state, np.float64).reshape(1, -1))
            self._queued_state = np.array(next_state, np.float64).reshape(-1)


class NAF(BaseAgent):
    def __init__(self, config):
        network_manager = NAF_Network_Manager(config)
        super(NAF, self).__init__(config, network_manager)
