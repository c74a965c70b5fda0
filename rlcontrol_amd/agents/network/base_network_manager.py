"""Base of the network managers (the object ``BaseAgent`` delegates to; agents/network/base_network_manager.py of
the reference).  It carries what every manager shares -- environment facts copied from the Config, the
train / eval step and episode counters, the exploration policy selected by ``config.exploration_policy`` and the
inert ``RunningMeanStd(state_dim)`` of quirk Q6 -- and leaves ``take_action`` / ``update_network`` to the
algorithm.  There is no tf.Graph here: the networks live in the HIP handle the subclass creates.
"""
from ...utils.exploration_policy import make_policy
from ...utils.running_mean_std import RunningMeanStd

_ENV_FACTS = ("state_dim", "state_min", "state_max", "action_dim", "action_min", "action_max")
_LOG_FLAGS = ("write_log", "write_plot", "writer")
_COUNTERS = ("train_global_steps", "eval_global_steps", "train_ep_count", "eval_ep_count")


def check_norm_type(config, agent_name, supported):
    """The reference accepts norm_type none / input_norm / layer / batch from any json (base_network.py:53-65).  The HIP
    path implements a subset per agent; anything else is refused loudly (never silently run as input_norm)."""
    if config.norm_type not in supported:
        raise ValueError("%s: norm_type %r is not implemented in the HIP path (implemented: %s)" %
                         (agent_name, config.norm_type, ", ".join(repr(s) for s in supported)))


class BaseNetwork_Manager(object):
    def __init__(self, config):
        self.random_seed = config.random_seed
        for name in _ENV_FACTS + _LOG_FLAGS:
            setattr(self, name, getattr(config, name))
        for name in _COUNTERS:
            setattr(self, name, 0)
        self.set_exploration(config)
        # 'input_norm', 'layer' and 'batch' all keep a RunningMeanStd around; only 'none' does not
        self.input_norm = None if config.norm_type == 'none' else RunningMeanStd(self.state_dim)

    def set_exploration(self, config):
        self.use_external_exploration, self.exploration_policy = make_policy(
            config, self.random_seed, self.action_dim, self.action_min, self.action_max)

    # ---- what the algorithm provides -------------------------------------------------------------
    def take_action(self, state, is_train, is_start):
        raise NotImplementedError

    def update_network(self, state_batch, action_batch, next_state_batch, reward_batch, gamma_batch):
        raise NotImplementedError

    # ---- between episodes (BaseAgent.reset) ------------------------------------------------------
    def reset(self):
        self.train_ep_count = self.eval_ep_count = 0
        if self.exploration_policy:
            self.exploration_policy.reset()
