#!/usr/bin/env python3
"""Throughput of the ReverseKL / ForwardKL fused update (jsonfiles/agent/reverse_kl.json shape: S=3, A=1, four 200-wide
layers, batch 32, N_param 64 -> 1984 (state, node) rows per update), 256 co-resident agents, device sampler and device
eps; the torch oracle timed beside it on the host.  Writes gpurun_out/<tag>_kl_bench.json.
    python scripts/bench_kl.py --tag r02"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DIMS = (3, 1, 200, 200, 200, 200)
# multiply-adds x 2 of one update at batch B with K nodes: three nets forward (+ Q at the new action), the K-node Q
# forward, two backward contractions and two weight-gradient contractions per hidden layer pair
def flop_per_update(B, K, S=3, L=200):
    fwd = lambda rows, k_in: 2 * rows * (k_in * L + L * L + L)
    return (fwd(B, S) * 3 + fwd(B, S + 1) * 2 + fwd(B * K, S + 1)        # pi, V, V' | Q(s,a), Q(s,a_new) | Q at the nodes
            + 3 * 2 * B * (L * L + L)                                    # input gradients of the three trained nets
            + 3 * 2 * B * (L * L + (S + 1) * L + 2 * L))                 # weight gradients


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=256)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--records", type=int, default=20000)
    ap.add_argument("--n_param", type=int, default=64)
    ap.add_argument("--updates", type=int, default=8)
    ap.add_argument("--tag", default="r02")
    ap.add_argument("--cpu_updates", type=int, default=300)
    a = ap.parse_args()
    from rlcontrol_amd import hip_kl
    rng = np.random.RandomState(0)
    n = a.records
    data = (rng.uniform(-1, 1, (n, 3)), rng.uniform(-2, 2, (n, 1)), rng.uniform(-16, 0, n), rng.uniform(-1, 1, (n, 3)),
            np.full(n, 0.99))
    out = {"agents": a.agents, "batch": a.batch, "n_param": a.n_param, "records_per_agent": n,
           "flop_per_update": flop_per_update(a.batch, a.n_param - 2)}
    for kind, kernel in (("reverse", "mfma"), ("forward", "mfma"), ("reverse", "generic")):
        pop = hip_kl.KLPopulation(kind, a.agents, *DIMS, a.batch, n, 0.01, 2.0, 1e-3, 1e-3, 0.1,
                                  seeds=np.arange(a.agents) + 1, n_param=a.n_param)
        pop.set_kernel(kernel)
        for i in range(a.agents):
            pop.set_params(i, hip_kl.init_params(*DIMS, i))
            pop.replay_add_batch(i, *data)
        pop.update(a.updates)
        pop.sync()
        pop.timer_begin()
        for _ in range(5):
            pop.update(a.updates)
        ms = pop.timer_end()
        rate = a.agents * a.updates * 5 / (ms * 1e-3)
        assert np.all(np.isfinite(pop.get_blob(0, "theta")))
        out[kind + "_" + kernel] = {"updates_per_s": rate, "ms_per_launch": ms / 5, "updates_per_launch": a.updates,
                                    "tflops": rate * out["flop_per_update"] / 1e12,
                                    "frac_of_fp32_matrix_peak": rate * out["flop_per_update"] / 157.3e12}
        print("%sKL %s: %.0f updates/s (%d agents), %.2f TFLOP/s" % (kind, kernel, rate, a.agents,
                                                                    out[kind + "_" + kernel]["tflops"]), flush=True)
        pop.close()
    if a.cpu_updates > 0:
        import torch
        from oracle import kl_torch as K
        d = K.KlDims(*DIMS)
        o = K.KLOracle("reverse", d, K.init_params(d, 0), 1e-3, 1e-3, 0.1, 0.01, 2.0, a.n_param)
        idx = [rng.choice(n, a.batch, replace=False) for _ in range(a.cpu_updates)]
        eps = rng.randn(a.cpu_updates, a.batch, 1)
        t0 = time.time()
        for k, j in enumerate(idx):
            o.update(data[0][j], data[1][j], data[3][j], data[2][j], data[4][j], eps[k])
        dt = time.time() - t0
        out["cpu_oracle"] = {"updates_per_s": a.cpu_updates / dt, "threads": torch.get_num_threads(), "sample": "%d updates" % a.cpu_updates}
        print("torch oracle on the host: %.0f updates/s (%d threads)" % (a.cpu_updates / dt, torch.get_num_threads()))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "%s_kl_bench.json" % a.tag)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
