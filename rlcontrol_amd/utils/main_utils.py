"""Agent factory and sweep indexing (mirrors utils/main_utils.py:7-99 of the reference).

``get_sweep_parameters`` is a mixed-radix decode of INDEX over the json ``sweeps`` dict in key order,
FIRST key varying fastest; the index wraps, so INDEX // n_settings is the run (= seed) number
(main.py:133-141).  ``create_agent`` maps the json ``"agent"`` string to a class; agents outside the
accelerated hot path are not provided and an unknown name prints and exits
with status 0 exactly like the reference (utils/main_utils.py:83-85).
"""
from collections import OrderedDict

_AGENTS = {
    "DDPG": ("rlcontrol_amd.agents.DDPG", "DDPG"),
    "SoftActorCritic": ("rlcontrol_amd.agents.SoftActorCritic", "SoftActorCritic"),
    "NAF": ("rlcontrol_amd.agents.NAF", "NAF"),
    "ReverseKL": ("rlcontrol_amd.agents.ReverseKL", "ReverseKL"),
    "ForwardKL": ("rlcontrol_amd.agents.ForwardKL", "ForwardKL"),
}


def create_agent(agent_string, config):
    entry = _AGENTS.get(agent_string)
    if entry is None:
        print("Don't know this agent")
        exit(0)
    import importlib
    module = importlib.import_module(entry[0])
    return getattr(module, entry[1])(config)


def get_sweep_parameters(parameters, index):
    chosen = OrderedDict()
    stride = 1
    for key, choices in parameters.items():
        count = len(choices)
        chosen[key] = choices[int(index / stride) % count]
        stride *= count
    return (chosen, stride)
