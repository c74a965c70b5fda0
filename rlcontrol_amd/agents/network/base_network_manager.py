"""Network-manager base (mirrors agents/network/base_network_manager.py:6-87 of the reference).

Holds the env facts, the step/episode counters, the exploration policy chosen from
``config.exploration_policy`` ('ou_noise' | 'epsilon_greedy' | 'random_uniform' | 'none', else
ValueError) and the inert ``RunningMeanStd(state_dim)`` (quirk Q6).  There is no tf.Graph: the
networks live in the HIP handle created by the subclass.
"""
from ...utils.running_mean_std import RunningMeanStd


class BaseNetwork_Manager(object):
    def __init__(self, config):
        self.random_seed = config.random_seed

        self.state_dim = config.state_dim
        self.state_min = config.state_min
        self.state_max = config.state_max

        self.action_dim = config.action_dim
        self.action_min = config.action_min
        self.action_max = config.action_max

        self.write_log = config.write_log
        self.write_plot = config.write_plot
        self.writer = config.writer

        self.train_global_steps = 0
        self.eval_global_steps = 0
        self.train_ep_count = 0
        self.eval_ep_count = 0

        self.use_external_exploration = None
        self.exploration_policy = None
        self.set_exploration(config)

        if config.norm_type != 'none':
            self.input_norm = RunningMeanStd(self.state_dim)
        else:
            self.input_norm = None

    def set_exploration(self, config):
        kind = config.exploration_policy
        if kind == 'ou_noise':
            from ...utils.exploration_policy import OrnsteinUhlenbeckProcess
            self.use_external_exploration = True
            self.exploration_policy = OrnsteinUhlenbeckProcess(
                self.random_seed, self.action_dim, self.action_min, self.action_max,
                theta=config.ou_theta, mu=config.ou_mu, sigma=config.ou_sigma)
        elif kind == 'epsilon_greedy':
            from ...utils.exploration_policy import EpsilonGreedy
            self.use_external_exploration = True
            self.exploration_policy = EpsilonGreedy(
                self.random_seed, self.action_min, self.action_max, config.annealing_steps,
                config.min_epsilon, config.max_epsilon, is_continuous=True)
        elif kind == 'random_uniform':
            from ...utils.exploration_policy import RandomUniform
            self.use_external_exploration = True
            self.exploration_policy = RandomUniform(self.random_seed, self.action_min, self.action_max,
                                                    is_continuous=True)
        elif kind == 'none':
            self.use_external_exploration = False
            self.exploration_policy = None
        else:
            raise ValueError("Invalid Value for config.exploration_policy")

    def take_action(self, state, is_train, is_start):
        raise NotImplementedError

    def update_network(self, state_batch, action_batch, next_state_batch, reward_batch, gamma_batch):
        raise NotImplementedError

    def reset(self):
        self.train_ep_count = 0
        self.eval_ep_count = 0
        if self.exploration_policy:
            self.exploration_policy.reset()
