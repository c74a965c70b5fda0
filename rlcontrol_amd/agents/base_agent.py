"""Agent interface (mirrors agents/base_agent.py:7-74 of the reference).

``start / step / update / reset`` are what Experiment calls (experiment.py:105-135,198-212).
Behaviour kept from the reference:
  * a truncated transition is NOT stored, a terminal one is stored with gamma 0.0, and ``learn()``
    runs after every ``update`` regardless (agents/base_agent.py:54-63, quirk Q7);
  * ``learn()`` fires only when ``size > max(warmup_steps, batch_size)`` -- strict (quirk Q12);
  * ``take_action`` raises NotImplementedError while ``size < warmup_steps`` (agents/base_agent.py:42-46);
  * ``input_norm.update`` is called every step although it never reaches the network (quirk Q6).
The one structural difference: sampling + gather + update_network run as ONE fused HIP launch
(``network_manager.update_from_replay``) instead of a host gather followed by seven Session.run calls.
"""
import numpy as np

from ..utils.replaybuffer import ReplayBuffer


class BaseAgent(object):
    def __init__(self, config, network_manager):
        self.norm_type = config.norm_type

        self.state_dim = config.state_dim
        self.state_min = config.state_min
        self.state_max = config.state_max

        self.action_dim = config.action_dim
        self.action_min = config.action_min
        self.action_max = config.action_max

        self.network_manager = network_manager
        self.replay_buffer = ReplayBuffer(config.buffer_size, config.random_seed,
                                          store=network_manager.device_replay(),
                                          sampler=getattr(config, "replay_sampler", "reference"))
        self.batch_size = config.batch_size
        self.warmup_steps = config.warmup_steps
        self.gamma = config.gamma

        self.write_log = config.write_log
        self.write_plot = config.write_plot
        self.writer = config.writer
        self.config = config

    def start(self, state, is_train):
        return self.take_action(state, is_train, is_start=True)

    def step(self, state, is_train):
        return self.take_action(state, is_train, is_start=False)

    def take_action(self, state, is_train, is_start):
        if self.replay_buffer.get_size() < self.warmup_steps:
            raise NotImplementedError
        return self.network_manager.take_action(state, is_train, is_start)

    def get_value(self, s, a):
        raise NotImplementedError

    def update(self, state, next_state, reward, action, is_terminal, is_truncated):
        if not is_truncated:
            gamma_i = 0.0 if is_terminal else self.gamma
            self.replay_buffer.add(state, action, reward, next_state, gamma_i)
        if self.norm_type != 'none':
            self.network_manager.input_norm.update(np.array([state]))
        self.learn()

    def learn(self):
        if self.replay_buffer.get_size() > max(self.warmup_steps, self.batch_size):
            idx = self.replay_buffer.sample_indices(self.batch_size)
            self.network_manager.update_from_replay(idx)

    def reset(self):
        self.network_manager.reset()
