#!/usr/bin/env python3
"""A whole INDEX sweep of Pendulum-v0 DDPG runs on ONE MI355X with the environment on the device
(main.py --device_rollout semantics).  Writes profiles/<tag>_device_sweep.json: wall-clock, steps/s, and per
index the evaluation curve summary.
    python scripts/device_sweep.py --indices 0 49 12544 --agent_json jsonfiles/agent/ddpg_b100.json --tag r01
(indices START STEP STOP as in main.py: STEP = 49 walks the seeds of setting 0, actor_lr 1e-3 / critic_lr 1e-2)"""
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
    ap.add_argument("--indices", type=int, nargs=3, default=[0, 49, 49 * 256])
    ap.add_argument("--agent_json", default=os.path.join(ROOT, "jsonfiles/agent/ddpg_b100.json"))
    ap.add_argument("--env_json", default=os.path.join(ROOT, "jsonfiles/environment/Pendulum-v0.json"))
    ap.add_argument("--total_mil_steps", type=float, default=None)
    ap.add_argument("--buffer_size", type=int, default=None, help="override (a run never stores more than its steps)")
    ap.add_argument("--tag", default="r01")
    args = ap.parse_args()
    with open(args.env_json) as f:
        env_json = json.load(f, object_pairs_hook=OrderedDict)
    if args.total_mil_steps is not None:
        env_json["TotalMilSteps"] = args.total_mil_steps
    with open(args.agent_json) as f:
        agent_json = json.load(f, object_pairs_hook=OrderedDict)
    if args.buffer_size is not None:
        agent_json["sweeps"]["buffer_size"] = [args.buffer_size]
    env = envs.create_environment(env_json)
    env_params = {"env_name": env.name, "state_dim": env.state_dim, "state_min": env.state_min,
                  "state_max": env.state_max, "action_dim": env.action_dim, "action_min": env.action_min,
                  "action_max": env.action_max}
    indices = list(range(args.indices[0], args.indices[2], args.indices[1]))
    data = drv.new_data_dict(agent_json, env_json)
    t0 = time.time()
    last = [t0]

    def progress(steps):
        if time.time() - last[0] > 20:
            last[0] = time.time()
            print("  %d steps/agent after %.0f s" % (steps, time.time() - t0), flush=True)

    drv.run_indices_on_device(indices, agent_json, env_json, env_params,
                              {"write_log": False, "write_plot": False, "device": 0}, data, verbose=True,
                              progress=progress)
    wall = time.time() - t0
    total = int(env_json["TotalMilSteps"] * 1000000)
    out = {"indices": args.indices, "n_runs": len(indices), "steps_per_run": total, "wall_s": wall,
           "env_steps_per_s": len(indices) * total / wall, "agent_json": os.path.basename(args.agent_json),
           "env": env_json, "settings": {}}
    for sweep, sd in data["experiment_data"].items():
        ev = np.array([r["eval_episode_rewards"].mean(1) for r in sd["runs"]])       # [runs][evals]
        last10 = ev[:, -10:].mean(1)
        out["settings"][str(sweep)] = {
            "agent_params": {k: v for k, v in sd["agent_params"].items() if k != "writer"},
            "n_runs": len(sd["runs"]), "first_eval_mean": float(ev[:, 0].mean()),
            "last10_mean": float(last10.mean()), "last10_stderr": float(last10.std() / np.sqrt(len(last10))),
            "curve_mean": ev.mean(0).tolist()}
        print("setting %s (%d runs): first eval %.1f -> last-10-evals %.1f +- %.1f" % (
            sweep, len(sd["runs"]), ev[:, 0].mean(), last10.mean(), last10.std() / np.sqrt(len(last10))))
    print("%d runs x %d steps in %.1f s  (%.0f env steps/s, %.1f us per population step)" % (
        len(indices), total, wall, out["env_steps_per_s"], 1e6 * wall / total))
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    path = os.path.join(ROOT, "profiles", "%s_device_sweep.json" % args.tag)
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path)


if __name__ == "__main__":
    run()
