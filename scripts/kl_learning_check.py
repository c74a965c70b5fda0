#!/usr/bin/env python3
"""Learning check of the ForwardKL / ReverseKL drop-in agents against the only reference-produced numbers for them:
the notebook output plots.ipynb:90 of the reference (exp.get_best_hp(data, "eval", after=0): mean evaluation return over
ALL 201 evaluations of a 100 k-step Pendulum-v0 run, averaged over runs) -- ForwardKL settings 18, 9, 0, 10, 19 of
jsonfiles/agent/forward_kl.json at -174.7, -178.0, -180.1, -181.0, -181.2; ReverseKL's best setting 1 at -342.8.
Runs main.py (host loop, one agent per process, up to --procs processes at a time) and writes
gpurun_out/<tag>_kl_learning.json.
    python scripts/kl_learning_check.py --tag r02 --agent forward_kl --indices 18 9 0 10 19"""
import argparse
import json
import os
import pickle
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = {"forward_kl": {18: -174.7213731458829, 9: -178.02861992504089, 0: -180.1086606262482, 10: -181.04537472115973,
                            19: -181.18048436377856},
             "reverse_kl": {1: -342.7578240532044, 2: -431.70688599980605, 5: -457.9279849866135, 4: -474.6405942152839}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r02")
    ap.add_argument("--agent", default="forward_kl", choices=sorted(REFERENCE))
    ap.add_argument("--indices", type=int, nargs="+", default=[18, 9, 0, 10, 19])
    ap.add_argument("--procs", type=int, default=5)
    a = ap.parse_args()
    out_root = os.path.join(ROOT, "gpurun_out", "%s_%s_learning" % (a.tag, a.agent))
    os.makedirs(out_root, exist_ok=True)
    pending, running, t0 = list(a.indices), [], time.time()
    while pending or running:
        while pending and len(running) < a.procs:
            i = pending.pop(0)
            d = os.path.join(out_root, "idx%d" % i)
            os.makedirs(d, exist_ok=True)
            log = open(os.path.join(d, "log.txt"), "w")
            p = subprocess.Popen([sys.executable, os.path.join(ROOT, "main.py"), "--env_json",
                                  os.path.join(ROOT, "jsonfiles/environment/Pendulum-v0.json"), "--agent_json",
                                  os.path.join(ROOT, "jsonfiles/agent/%s.json" % a.agent), "--indices", str(i), "1", str(i + 1),
                                  "--save_dir", d, "--quiet"], stdout=log, stderr=subprocess.STDOUT, cwd=ROOT)
            running.append((i, p, d))
        time.sleep(20)
        print("[%5.0f s] running %s, pending %s" % (time.time() - t0, [r[0] for r in running], pending), flush=True)
        for r in list(running):
            if r[1].poll() is not None:
                running.remove(r)
                if r[1].returncode != 0:
                    print("index %d failed (rc %d), see %s/log.txt" % (r[0], r[1].returncode, r[2]), flush=True)
    res = {"agent": a.agent, "env": "Pendulum-v0, 100 k steps, 201 evaluations x 10 episodes", "seeds": 1, "settings": {}}
    for i in a.indices:
        d = os.path.join(out_root, "idx%d" % i)
        pk = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith(".pkl")]
        if not pk:
            continue
        with open(pk[0], "rb") as f:
            data = pickle.load(f)                    # written by this repo's main.py in this very run
        for hp, entry in data["experiment_data"].items():
            ev = np.stack([np.asarray(run["eval_episode_rewards"]) for run in entry["runs"]]).mean(axis=-1)
            res["settings"][int(hp)] = {"mean_eval_return_all_evals": float(ev.mean()),
                                        "mean_eval_return_last_10": float(ev[:, -10:].mean()),
                                        "reference_plots_ipynb_90": REFERENCE[a.agent].get(int(hp)),
                                        "agent_params": {k: v for k, v in entry["agent_params"].items()
                                                         if k in ("pi_lr", "qf_vf_lr", "entropy_scale")}}
    res["wall_s"] = time.time() - t0
    path = os.path.join(ROOT, "gpurun_out", "%s_%s_learning.json" % (a.tag, a.agent))
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
