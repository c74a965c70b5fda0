#!/usr/bin/env python3
"""Throughput of the SAC-v1 and NAF update kernels (parity paths, fp32 VALU generic kernels) at the shapes of
BASELINE.json configs[2] / configs[3] (SAC S=3, A=1, L=128; NAF S=8, A=2, L=200; synthetic
U-distributed replay), 256 co-resident agents, device sampler.  Writes profiles/<tag>_sac_naf_bench.json.
    python scripts/bench_sac_naf.py --tag r01"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fill(pop, S, A, n, rng):
    data = (rng.uniform(-1, 1, (n, S)), rng.uniform(-1, 1, (n, A)), rng.uniform(-16, 0, n), rng.uniform(-1, 1, (n, S)),
            np.full(n, 0.99))
    for i in range(pop.n_agents):
        pop.replay_add_batch(i, *data)


def timed(pop, updates, launches):
    pop.update(updates)
    pop.sync()
    pop.timer_begin()
    for _ in range(launches):
        pop.update(updates)
    ms = pop.timer_end()
    return pop.n_agents * updates * launches / (ms * 1e-3), ms / launches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=256)
    ap.add_argument("--batch", type=int, default=100)
    ap.add_argument("--records", type=int, default=20000)
    ap.add_argument("--tag", default="r02")
    ap.add_argument("--kernel", default="auto", choices=["auto", "generic", "mfma"])
    ap.add_argument("--updates", type=int, default=8)
    ap.add_argument("--only", default="", choices=["", "sac", "naf"])
    a = ap.parse_args()
    from rlcontrol_amd import hip_naf, hip_sac
    rng = np.random.RandomState(0)
    out = {"agents": a.agents, "batch": a.batch, "records_per_agent": a.records}
    NA = a.agents
    sac = hip_sac.SACPopulation(NA, 3, 1, 128, 128, 128, 128, a.batch, a.records, 0.01, -8.0, 8.0, 2.0, 1e-3, 1e-3, 0.2,
                                seeds=np.arange(NA) + 1)
    for i in range(NA):
        sac.set_params(i, hip_sac.init_params(3, 1, 128, 128, 128, 128, i))
    if a.kernel != "auto":
        sac.set_kernel(a.kernel)
    if a.only in ("", "sac"):
        fill(sac, 3, 1, a.records, rng)
        rate, ms = timed(sac, a.updates, 5)
        out["sac"] = {"updates_per_s": rate, "ms_per_launch": ms, "updates_per_launch": a.updates, "kernel": sac.kernel_in_use(),
                      "shape": "S=3 A=1 L=128 (jsonfiles/agent/sac.json)"}
        print("SAC-v1: %.0f updates/s (%d agents, %s kernel)" % (rate, NA, sac.kernel_in_use()), flush=True)
    sac.close()
    if a.only == "sac":
        print(json.dumps(out))
        return
    naf = hip_naf.NAFPopulation(NA, 8, 2, 200, 200, a.batch, a.records, 0.01, -np.ones(8) * 10, np.ones(8) * 10,
                                np.ones(2), 1e-3, seeds=np.arange(NA) + 1)
    for i in range(NA):
        naf.set_params(i, hip_naf.init_params(8, 2, 200, 200, i))
    if a.kernel != "auto":
        naf.set_kernel(a.kernel)
    fill(naf, 8, 2, a.records, rng)
    rate, ms = timed(naf, a.updates, 5)
    out["naf"] = {"updates_per_s": rate, "ms_per_launch": ms, "updates_per_launch": a.updates, "kernel": naf.kernel_in_use(),
                  "shape": "S=8 A=2 L=200 (jsonfiles/agent/naf.json)"}
    print("NAF: %.0f updates/s (%d agents, %s kernel)" % (rate, NA, naf.kernel_in_use()), flush=True)
    naf.close()
    path = os.path.join(ROOT, "gpurun_out", "%s_sac_naf_bench_%s.json" % (a.tag, a.kernel))
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
