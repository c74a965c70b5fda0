"""Run configuration object (mirrors the reference's utils/config.py:1-27).

Defaults are the reference's: warmup 0, batch 32, buffer 1e6, tau 0.01, gamma 0.99,
OU theta/mu/sigma 0.15/0.0/0.2; ``norm`` and ``exploration_policy`` start as None.
``merge_config`` lets ANY key of the env facts, the json sweep or the CLI override or add an
attribute (utils/config.py:24-27) -- e.g. ``"batch_size": [100]`` in a sweep.
"""


class Config(object):
    _DEFAULTS = (
        ("norm", None),
        ("exploration_policy", None),
        ("warmup_steps", 0),
        ("batch_size", 32),
        ("buffer_size", 1e6),
        ("tau", 0.01),
        ("gamma", 0.99),
        ("ou_theta", 0.15),
        ("ou_mu", 0.0),
        ("ou_sigma", 0.2),
    )

    def __init__(self):
        for key, value in self._DEFAULTS:
            setattr(self, key, value)

    def merge_config(self, custom_config):
        for key, value in custom_config.items():
            setattr(self, key, value)
