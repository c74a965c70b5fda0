#!/usr/bin/env python3
"""Single-agent drop-in path: wall time per environment step (replay_add + fused update + act round trip)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_agent import _config  # noqa
from rlcontrol_amd.utils.main_utils import create_agent

for batch in (32, 100):
    cfg, env = _config(0, batch)
    agent = create_agent("DDPG", cfg)
    env.set_random_seed(0)
    obs = env.reset(); agent.reset(); a = agent.start(obs, True)
    times = {"update": 0.0, "act": 0.0, "env": 0.0}
    n = 0
    for t in range(1500):
        t0 = time.perf_counter(); obs_n, r, done, _ = env.step(a); t1 = time.perf_counter()
        agent.update(obs, obs_n, float(r), a, done, bool(done)); t2 = time.perf_counter()
        if done:
            obs = env.reset(); agent.reset(); a = agent.start(obs, True)
        else:
            a = agent.step(obs_n, True); obs = obs_n
        t3 = time.perf_counter()
        if t > 300:
            times["env"] += t1 - t0; times["update"] += t2 - t1; times["act"] += t3 - t2; n += 1
    print("batch %d: per step  env %.1f us  update(add+sample+launch) %.1f us  act(round trip) %.1f us" % (
        batch, 1e6 * times["env"] / n, 1e6 * times["update"] / n, 1e6 * times["act"] / n))
