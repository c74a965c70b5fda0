"""Environment wrapper (mirrors environments/environments.py:16-156 of the reference).

``create_environment(env_json)`` returns an object with the attributes Experiment and main.py read:
``name, eval_interval, eval_episodes, TOTAL_STEPS_LIMIT, EPISODE_STEPS_LIMIT, state_dim/min/max,
action_dim/min/max`` and ``set_random_seed / reset / step / close``.  Pendulum-v0 is served by the
in-tree restatement (gym is absent here); any other name is looked up in an installed ``gym`` and
fails loudly when there is none.  The reference's Bimodal toy environments are out of scope
(SURVEY.md section 2, row 14).
"""
import numpy as np

from .pendulum import PendulumEnv


def _make_instance(name):
    if name == 'Pendulum-v0':
        return PendulumEnv()
    try:
        import gym  # noqa: F401
    except ImportError:
        raise RuntimeError("environment %r needs gym, which is not installed; only Pendulum-v0 is built in" % name)
    return gym.make(name)


def create_environment(env_params):
    return ContinuousEnvironment(env_params)


class ContinuousEnvironment(object):
    def __init__(self, env_params):
        self.name = env_params['environment']
        self.eval_interval = env_params['EvalIntervalMilSteps'] * 1000000
        self.eval_episodes = env_params['EvalEpisodes']
        self.instance = _make_instance(self.name)

        self.TOTAL_STEPS_LIMIT = env_params['TotalMilSteps'] * 1000000
        if env_params['EpisodeSteps'] != -1:
            self.EPISODE_STEPS_LIMIT = env_params['EpisodeSteps']
            self.instance._max_episode_steps = env_params['EpisodeSteps']
        else:
            self.EPISODE_STEPS_LIMIT = self.instance._max_episode_steps

        obs, act = self.instance.observation_space, self.instance.action_space
        self.state_dim = obs.shape[0]
        self.state_range = obs.high - obs.low
        self.state_min = obs.low
        self.state_max = obs.high
        self.state_bounded = not (np.any(np.isinf(obs.high)) or np.any(np.isinf(obs.low)))

        self.action_dim = int(act.sample().shape[0])
        self.action_range = act.high - act.low
        self.action_min = act.low
        self.action_max = act.high

    def set_random_seed(self, random_seed):
        self.instance.seed(random_seed)

    def reset(self):
        return self.instance.reset()

    def step(self, action):
        return self.instance.step(action)

    def close(self):
        self.instance.close()
