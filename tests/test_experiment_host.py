"""Experiment loop / driver semantics (CPU only, scripted fake agent): eval interleave, truncation rule,
return tuple, result-pickle schema (experiment.py:48-217, main.py:80-95,188-209 of the reference)."""
import json
import os
import pickle

import numpy as np

from rlcontrol_amd.environments.environments import create_environment
from rlcontrol_amd.experiment import Experiment

ENV = {"environment": "Pendulum-v0", "TotalMilSteps": 0.00045, "EpisodeSteps": -1,
       "EvalIntervalMilSteps": 0.0001, "EvalEpisodes": 2}


class ScriptAgent(object):
    def __init__(self):
        self.log = []

    def start(self, s, is_train):
        self.log.append(("start", is_train))
        return np.array([0.5])

    def step(self, s, is_train):
        self.log.append(("step", is_train))
        return np.array([-0.5])

    def update(self, s, s2, r, a, term, trunc):
        assert isinstance(r, float)
        self.log.append(("update", term, trunc))

    def reset(self):
        self.log.append(("reset",))


def test_experiment_schedule_and_truncation():
    agent = ScriptAgent()
    exp = Experiment(agent, create_environment(ENV), create_environment(ENV), seed=0, verbose=False)
    out = exp.run()
    (train_rewards, eval_rewards, train_steps, eval_steps, ts_at_eval, train_time, eval_time, n_train_eps,
     cum_steps) = out
    # 450 total steps, eval every 100 training steps and once at step 0 -> evals at 0,100,200,300,400
    assert ts_at_eval == [0, 100, 200, 300, 400]
    assert np.array(eval_rewards).shape == (5, 2) and np.array(eval_steps).tolist() == [[200, 200]] * 5
    # episodes of 200 steps (TimeLimit): two complete, the third cut by TOTAL_STEPS_LIMIT is not recorded
    assert train_steps == [200, 200] and n_train_eps == 3 and cum_steps == [200, 400]
    updates = [e for e in agent.log if e[0] == "update"]
    assert len(updates) == 450
    # the 200th step of each complete episode is done AND truncated (experiment.py:127-128): update still called
    trunc = [i for i, e in enumerate(updates) if e[2]]
    assert trunc == [199, 399]
    assert all(e[1] for i, e in enumerate(updates) if i in trunc)
    # every eval episode resets the agent; evals happen inside training episodes (Q8)
    assert sum(1 for e in agent.log if e == ("reset",)) == 3 + 5 * 2
    assert sum(1 for e in agent.log if e == ("start", False)) == 10


def test_pendulum_matches_gym_definition():
    env = create_environment(ENV)
    env.set_random_seed(3)
    s = env.reset()
    th, thd = env.instance.state
    assert np.allclose(s, [np.cos(th), np.sin(th), thd]) and abs(th) <= np.pi and abs(thd) <= 1.0
    s2, r, done, _ = env.step(np.array([5.0]))          # clipped to max torque 2
    wrapped = ((th + np.pi) % (2 * np.pi)) - np.pi
    assert np.isclose(r, -(wrapped ** 2 + 0.1 * thd ** 2 + 0.001 * 4.0))
    thd2 = np.clip(thd + (-15.0 * np.sin(th + np.pi) + 3.0 * 2.0) * 0.05, -8, 8)
    assert np.allclose(env.instance.state, [th + (thd + (-15.0 * np.sin(th + np.pi) + 6.0) * 0.05) * 0.05, thd2])
    assert not done and env.EPISODE_STEPS_LIMIT == 200
    assert env.state_dim == 3 and env.action_dim == 1 and env.action_max.tolist() == [2.0]
    assert env.state_max.tolist() == [1.0, 1.0, 8.0]
    env2 = create_environment(ENV)
    env2.set_random_seed(3)
    assert np.array_equal(env2.reset(), s)               # seeded resets are reproducible


def test_main_result_pickle_schema(tmp_path, monkeypatch):
    import main as drv

    class FakeAgentCls(ScriptAgent):
        def __init__(self, config):
            ScriptAgent.__init__(self)
            assert config.batch_size == 32 and config.actor_lr in (0.001, 0.005) and config.random_seed == 1
            assert config.state_dim == 3 and config.exploration_policy == "ou_noise" and config.writer is None

    monkeypatch.setattr(drv, "create_agent", lambda name, cfg: FakeAgentCls(cfg))
    envf = tmp_path / "Pendulum-v0.json"
    envf.write_text(json.dumps(ENV))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data = drv.main(["--env_json", str(envf), "--agent_json", os.path.join(root, "jsonfiles/agent/ddpg.json"),
                     "--indices", "49", "1", "51", "--save_dir", str(tmp_path / "res"), "--quiet"])
    path = tmp_path / "res" / "Pendulum-v0_ddpgresults" / "data_49_1_51.pkl"
    with open(path, "rb") as f:
        disk = pickle.load(f)
    assert sorted(disk["experiment_data"].keys()) == [0, 1] == sorted(data["experiment_data"].keys())
    run = disk["experiment_data"][0]["runs"][0]
    assert run["random_seed"] == 1                        # index 49 // 49 settings
    for key in ("total_timesteps", "eval_interval_timesteps", "episodes_per_eval", "eval_episode_rewards",
                "eval_episode_steps", "timesteps_at_eval", "train_episode_steps", "train_episode_rewards",
                "total_train_episodes", "eval_time", "train_time"):
        assert key in run
    assert run["eval_episode_rewards"].shape == (5, 2)
    assert disk["experiment"]["agent"]["agent_name"] == "DDPG"
    assert disk["experiment_data"][1]["agent_params"]["actor_lr"] == 0.005


def test_presampled_indices_follow_the_reference_stream():
    """ReplayBuffer.presample draws the next minibatch early; a wrong guess of the buffer size (a truncated step stores
    nothing) is undone: the index stream equals the plain sampler's, call for call."""
    import numpy as np
    from rlcontrol_amd.utils.replaybuffer import ReplayBuffer

    class _Store(object):
        def __init__(self):
            self.n = 0

        def replay_size(self, agent):
            return self.n

    plain_store, pre_store = _Store(), _Store()
    plain = ReplayBuffer(10000, 5, store=(plain_store, 0))
    pre = ReplayBuffer(10000, 5, store=(pre_store, 0))
    rng = np.random.RandomState(0)
    size = 40
    for step in range(300):
        stored = rng.rand() > 0.15                  # 15 % of the steps are "truncated": nothing is stored
        size += 1 if stored else 0
        plain_store.n = pre_store.n = size
        a = plain.sample_indices(32)
        b = pre.sample_indices(32)
        assert np.array_equal(a, b), step
        pre.presample(32, size + 1)                 # always guesses that the next transition will be stored
