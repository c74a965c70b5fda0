#!/usr/bin/env python3
"""Pendulum-v0 DDPG learning curves through the drop-in path (main.py semantics, INDEX = seed*n_settings).
Writes profiles/<tag>_pendulum_curve.json: per seed the 201 x 10 eval returns (reference schema fields).
    python scripts/learning_curve.py --seeds 4 --agent_json jsonfiles/agent/ddpg_b100.json --tag r01"""
import argparse
import json
import os
import sys
import time
from collections import OrderedDict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import main as drv  # noqa: E402
import rlcontrol_amd.environments.environments as envs  # noqa: E402


def run():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=4)
    ap.add_argument("--agent_json", default=os.path.join(ROOT, "jsonfiles/agent/ddpg_b100.json"))
    ap.add_argument("--env_json", default=os.path.join(ROOT, "jsonfiles/environment/Pendulum-v0.json"))
    ap.add_argument("--total_mil_steps", type=float, default=None)
    ap.add_argument("--tag", default="r01")
    args = ap.parse_args()
    with open(args.env_json) as f:
        env_json = json.load(f, object_pairs_hook=OrderedDict)
    if args.total_mil_steps is not None:
        env_json["TotalMilSteps"] = args.total_mil_steps
    with open(args.agent_json) as f:
        agent_json = json.load(f, object_pairs_hook=OrderedDict)
    train_env, test_env = envs.create_environment(env_json), envs.create_environment(env_json)
    env_params = {"env_name": train_env.name, "state_dim": train_env.state_dim, "state_min": train_env.state_min,
                  "state_max": train_env.state_max, "action_dim": train_env.action_dim,
                  "action_min": train_env.action_min, "action_max": train_env.action_max}
    n_settings = drv.get_sweep_parameters(agent_json["sweeps"], 0)[1]
    data = drv.new_data_dict(agent_json, env_json)
    out = {"agent_json": os.path.basename(args.agent_json), "env": env_json, "runs": []}
    for seed in range(args.seeds):
        t0 = time.time()
        rd = drv.run_index(seed * n_settings, agent_json, env_json, train_env, test_env, env_params,
                           {"write_log": False, "write_plot": False, "device": 0}, data, verbose=False)
        ev = np.asarray(rd["eval_episode_rewards"])
        out["runs"].append({"seed": seed, "wall_s": time.time() - t0, "timesteps_at_eval": np.asarray(rd["timesteps_at_eval"]).tolist(),
                            "eval_mean": ev.mean(1).tolist(), "eval_returns": ev.tolist(),
                            "train_episode_rewards": np.asarray(rd["train_episode_rewards"]).tolist()})
        print("seed %d: %.0f s, eval return first %.1f -> last-10-evals mean %.1f (best eval %.1f)" % (
            seed, time.time() - t0, ev[0].mean(), ev[-10:].mean(), ev.mean(1).max()), flush=True)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    path = os.path.join(ROOT, "profiles", "%s_pendulum_curve.json" % args.tag)
    with open(path, "w") as f:
        json.dump(out, f)
    last = np.array([np.mean(r["eval_mean"][-10:]) for r in out["runs"]])
    print("last-10-evals mean return over %d seeds: %.1f +- %.1f (stderr)" % (len(last), last.mean(), last.std() / np.sqrt(len(last))))
    print("wrote", path)


if __name__ == "__main__":
    run()
