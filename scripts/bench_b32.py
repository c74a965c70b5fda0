#!/usr/bin/env python3
"""DDPG / SAC / NAF update throughput at the shipped jsons' batch 32 (256 agents and one agent), device sampler.
    RLCONTROL_HIP_LIB=ab/x.so python scripts/bench_b32.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(pop, updates, launches):
    pop.update(updates)
    pop.sync()
    pop.timer_begin()
    for _ in range(launches):
        pop.update(updates)
    ms = pop.timer_end()
    return pop.n_agents * updates * launches / (ms * 1e-3), ms * 1e3 / (launches * updates)


def main():
    from rlcontrol_amd import hip_ddpg, hip_sac, hip_naf
    B, N = int(os.environ.get("B", "32")), 20000
    rng = np.random.RandomState(0)
    for NA in (256, 1):
        d3 = (rng.uniform(-1, 1, (N, 3)), rng.uniform(-1, 1, (N, 1)), rng.uniform(-16, 0, N), rng.uniform(-1, 1, (N, 3)), np.full(N, 0.99))
        d8 = (rng.uniform(-1, 1, (N, 8)), rng.uniform(-1, 1, (N, 2)), rng.uniform(-16, 0, N), rng.uniform(-1, 1, (N, 8)), np.full(N, 0.99))
        pops = [
            ("DDPG", hip_ddpg.DDPGPopulation(NA, 3, 1, 200, 200, 200, B, N, 0.01, [-1, -1, -8], [1, 1, 8], [-2.0], [2.0], 1e-3, 1e-2,
                                             seeds=np.arange(NA) + 1), lambda i: hip_ddpg.init_params(3, 1, 200, 200, 200, i), d3),
            ("SAC", hip_sac.SACPopulation(NA, 3, 1, 128, 128, 128, 128, B, N, 0.01, -8.0, 8.0, 2.0, 1e-3, 1e-3, 0.1,
                                          seeds=np.arange(NA) + 1), lambda i: hip_sac.init_params(3, 1, 128, 128, 128, 128, i), d3),
            ("NAF", hip_naf.NAFPopulation(NA, 8, 2, 200, 200, B, N, 0.01, -np.ones(8), np.ones(8), np.ones(2), 1e-3,
                                          seeds=np.arange(NA) + 1), lambda i: hip_naf.init_params(8, 2, 200, 200, i), d8),
        ]
        for name, pop, init, data in pops:
            for i in range(NA):
                pop.set_params(i, init(i))
                pop.replay_add_batch(i, *data)
            rate, us = timed(pop, 16, 8)
            print("%-5s batch %d, %3d agents (%s): %9.0f updates/s, %6.1f us per update per agent" % (
                name, B, NA, pop.kernel_in_use(), rate, us), flush=True)
            pop.close()


if __name__ == "__main__":
    main()
