#!/usr/bin/env python3
"""ReverseKL / ForwardKL, ONE agent (and a few), shipped shape (batch 32, N_param 64): microseconds per update with the
node passes of the action integral over C workgroups (rlc_kl_set_split).  Writes gpurun_out/<tag>_kl_split_bench.json.
    python scripts/kl_latency.py --tag r02c"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DIMS = (3, 1, 200, 200, 200, 200)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r02c")
    ap.add_argument("--updates", type=int, default=64)
    a = ap.parse_args()
    from rlcontrol_amd import hip_kl
    rng = np.random.RandomState(0)
    n = 20000
    data = (rng.uniform(-1, 1, (n, 3)), rng.uniform(-2, 2, (n, 1)), rng.uniform(-16, 0, n), rng.uniform(-1, 1, (n, 3)),
            np.full(n, 0.99))
    out = {"batch": 32, "n_param": 64, "updates_per_launch": a.updates, "rows": []}
    for kind in ("reverse", "forward"):
        for agents, splits in ((1, (1, 2, 4, 8)), (8, (1, 4, 8)), (32, (1, 8))):
            for C in splits:
                pop = hip_kl.KLPopulation(kind, agents, *DIMS, 32, n, 0.01, 2.0, 1e-3, 1e-3, 0.1, seeds=np.arange(agents) + 1,
                                          n_param=64)
                pop.set_kernel("mfma")
                for i in range(agents):
                    pop.set_params(i, hip_kl.init_params(*DIMS, i))
                    pop.replay_add_batch(i, *data)
                pop.set_split(C)
                pop.update(a.updates)
                pop.sync()
                pop.timer_begin()
                for _ in range(4):
                    pop.update(a.updates)
                ms = pop.timer_end()
                us = ms * 1e3 / (4 * a.updates)
                out["rows"].append({"kind": kind, "agents": agents, "workgroups_per_agent": C, "us_per_update": us,
                                    "updates_per_s": agents * 4 * a.updates / (ms * 1e-3)})
                print("%sKL  %2d agents x %d workgroups: %7.1f us per update" % (kind, agents, C, us), flush=True)
                pop.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "%s_kl_split_bench.json" % a.tag)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
