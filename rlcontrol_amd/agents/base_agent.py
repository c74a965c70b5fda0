"""The agent object ``Experiment`` drives: ``start / step / update / reset`` (agents/base_agent.py of the reference;
call sites experiment.py:105-135,198-212).

Semantics kept, each covered by tests/test_golden_host.py against vectors generated from the reference:
  * insert rule -- a truncated transition is dropped, a terminal one is stored with transition gamma 0.0, any
    other with ``config.gamma``; ``learn()`` runs after every ``update`` whether or not something was stored (Q7);
  * learn gate -- strictly ``replay size > max(warmup_steps, batch_size)`` (Q12);
  * acting before ``warmup_steps`` transitions exist raises NotImplementedError (never with the default 0);
  * ``input_norm.update`` is fed every state although nothing reads it (Q6).
Structural difference: drawing the minibatch indices happens here (reference RNG stream), but gathering the
minibatch and the whole update run as one fused launch on the device replay (``update_from_replay``).
"""
import numpy as np

from ..utils.replaybuffer import ReplayBuffer

_COPIED = ("norm_type", "state_dim", "state_min", "state_max", "action_dim", "action_min", "action_max",
           "batch_size", "warmup_steps", "gamma", "write_log", "write_plot", "writer")


class BaseAgent(object):
    def __init__(self, config, network_manager):
        self.config = config
        for name in _COPIED:
            setattr(self, name, getattr(config, name))
        self.network_manager = network_manager
        self.replay_buffer = ReplayBuffer(config.buffer_size, config.random_seed,
                                          store=network_manager.device_replay(),
                                          sampler=getattr(config, "replay_sampler", "reference"))

    # ---- acting ----------------------------------------------------------------------------------
    def take_action(self, state, is_train, is_start):
        if self.replay_buffer.get_size() < self.warmup_steps:
            raise NotImplementedError          # the reference has no warm-up policy either
        return self.network_manager.take_action(state, is_train, is_start)

    def start(self, state, is_train):
        return self.take_action(state, is_train, is_start=True)

    def step(self, state, is_train):
        return self.take_action(state, is_train, is_start=False)

    def get_value(self, s, a):
        raise NotImplementedError

    # ---- learning --------------------------------------------------------------------------------
    def _transition_gamma(self, is_terminal):
        return 0.0 if is_terminal else self.gamma

    def update(self, state, next_state, reward, action, is_terminal, is_truncated):
        if not is_truncated:
            self.replay_buffer.add(state, action, reward, next_state, self._transition_gamma(is_terminal))
        # Experiment asks step(next_state) next unless the episode ended here (experiment.py:127-135)
        self.learn(None if (is_terminal or is_truncated) else next_state)
        # the running statistics nothing reads (Q6) are fed AFTER the update was launched: the reference feeds them
        # before learn() (agents/base_agent.py:60-62), but they touch neither the networks nor any random stream, and
        # here their ten microseconds of numpy then run while the GPU works
        if self.norm_type != 'none':
            self.network_manager.input_norm.update(np.array([state]))

    def learn(self, next_state=None):
        ready = self.replay_buffer.get_size() > max(self.warmup_steps, self.batch_size)
        if ready:
            indices = self.replay_buffer.sample_indices(self.batch_size)
            if next_state is not None and getattr(self.network_manager, "queues_next_action", False):
                self.network_manager.update_from_replay(indices, next_state=next_state)
            else:
                self.network_manager.update_from_replay(indices)
            # the next step's minibatch indices, drawn while the GPU is busy (the next transition is stored unless the
            # step is truncated: ReplayBuffer.presample undoes a wrong guess)
            if hasattr(self.replay_buffer, "presample"):
                self.replay_buffer.presample(self.batch_size, min(self.replay_buffer.get_size() + 1,
                                                                  int(self.replay_buffer.buffer_size)))

    def reset(self):
        self.network_manager.reset()
