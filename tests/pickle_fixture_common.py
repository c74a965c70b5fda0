"""Shared by tests/golden/make_pickle_golden.py (build container, imports the reference's consumers) and
tests/test_pickle_consumers.py (no reference): writes result pickles with THIS repo's main.py (reference schema,
main.py:80-95,188-209) driven by a deterministic scripted agent whose behaviour depends on the swept setting."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = {"environment": "Pendulum-v0", "TotalMilSteps": 0.0012, "EpisodeSteps": -1,
       "EvalIntervalMilSteps": 0.0004, "EvalEpisodes": 3}
# two invocations of main.py, as two SLURM array tasks would make them: settings 0..3 of run 0, settings 0..3 of run 1
INDEX_RANGES = (("0", "1", "4"), ("49", "1", "53"))


class SettingScriptAgent(object):
    """torque depends on the swept actor_lr (so that settings rank differently) and alternates in sign"""
    def __init__(self, config):
        self.amp = min(2.0, 200.0 * float(config.actor_lr))
        self.t = 0

    def start(self, s, is_train):
        self.t = 0
        return np.array([self.amp])

    def step(self, s, is_train):
        self.t += 1
        return np.array([self.amp if (self.t // 7) % 2 == 0 else -self.amp])

    def update(self, s, s2, r, a, term, trunc):
        pass

    def reset(self):
        pass


def write_pickles(out_dir):
    """run main.py twice; returns the paths of the two result pickles"""
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import main as drv
    os.makedirs(out_dir, exist_ok=True)
    envf = os.path.join(out_dir, "Pendulum-v0.json")
    with open(envf, "w") as f:
        json.dump(ENV, f)
    old = drv.create_agent
    drv.create_agent = lambda name, cfg: SettingScriptAgent(cfg)
    paths = []
    try:
        for rng in INDEX_RANGES:
            drv.main(["--env_json", envf, "--agent_json", os.path.join(ROOT, "jsonfiles/agent/ddpg.json"),
                      "--indices", rng[0], rng[1], rng[2], "--save_dir", os.path.join(out_dir, "res"), "--quiet"])
            paths.append(os.path.join(out_dir, "res", "Pendulum-v0_ddpgresults", "data_%s_%s_%s.pkl" % rng))
    finally:
        drv.create_agent = old
    return paths
