#!/usr/bin/env python3
"""Latency of ONE agent's DDPG update (the reference's deployment unit: one agent per process) on the fused MFMA
kernel (one CU) and in latency mode (rlc_ddpg_set_split: the minibatch over C CUs of one XCD), BASELINE configs[1]
shape (obs 3, act 1, 200/200/200, batch 100).  Also a few agents side by side.  Writes gpurun_out/<tag>_split_bench.json.
    python scripts/bench_split.py --tag r02"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r02")
    ap.add_argument("--batch", type=int, default=100)
    ap.add_argument("--records", type=int, default=100000)
    ap.add_argument("--updates", type=int, default=256)
    a = ap.parse_args()
    from rlcontrol_amd.hip_ddpg import DDPGPopulation, init_params
    rng = np.random.RandomState(0)
    n = a.records
    data = (rng.uniform(-1, 1, (n, 3)), rng.uniform(-2, 2, (n, 1)), rng.uniform(-16, 0, n), rng.uniform(-1, 1, (n, 3)),
            np.full(n, 0.99))
    out = {"batch": a.batch, "records_per_agent": n, "updates_per_launch": a.updates, "rows": []}
    for agents, splits in ((1, (1, 2, 4, 7, 8)), (8, (1, 4, 7)), (32, (1, 4, 8))):
        for C in splits:
            if a.batch > 64 * C and C > 1:
                continue
            pop = DDPGPopulation(agents, 3, 1, 200, 200, 200, a.batch, n, 0.01, [-1, -1, -8], [1, 1, 8], [-2.0], [2.0],
                                 1e-3, 1e-2, seeds=np.arange(agents) + 1)
            pop.set_kernel("mfma")
            pop.set_split(C)
            for i in range(agents):
                pop.set_params(i, init_params(3, 1, 200, 200, 200, i))
                pop.replay_add_batch(i, *data)
            pop.update(a.updates)
            pop.sync()
            pop.timer_begin()
            for _ in range(4):
                pop.update(a.updates)
            ms = pop.timer_end()
            us = ms * 1e3 / (4 * a.updates)
            assert np.all(np.isfinite(pop.get_blob(0, "theta")))
            row = {"agents": agents, "workgroups_per_agent": C, "us_per_update": us,
                   "updates_per_s": agents * 4 * a.updates / (ms * 1e-3)}
            out["rows"].append(row)
            print(row, flush=True)
            pop.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "%s_split_bench.json" % a.tag), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
