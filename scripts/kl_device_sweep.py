#!/usr/bin/env python3
"""The reference's whole forward_kl.json / reverse_kl.json hyper-parameter sweep x SEEDS on the device loop
(main.py --device_rollout: every INDEX is one agent of a population, environment simulated on the GPU), then the
table the reference's notebook prints for it (plots.ipynb:95-101: exp.get_best_hp(data, "eval", after=0) = mean
evaluation return over all evaluations and runs per setting, ascending).  Writes gpurun_out/<tag>_<agent>_device_sweep.json.
    python scripts/kl_device_sweep.py --agent forward_kl --seeds 5 --tag r02"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the rows the reference's notebook shows (setting, mean eval return); reverse_kl: NaN rows omitted
REFERENCE_TAIL = {"forward_kl": [(19, -181.18048436377856), (10, -181.04537472115973), (0, -180.1086606262482),
                                 (9, -178.02861992504089), (18, -174.7213731458829)],
                  "reverse_kl": [(4, -474.6405942152839), (5, -457.9279849866135), (2, -431.70688599980605),
                                 (1, -342.7578240532044)]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agent", default="forward_kl", choices=sorted(REFERENCE_TAIL))
    ap.add_argument("--seeds", type=int, default=5)
    ap.add_argument("--tag", default="r02")
    a = ap.parse_args()
    import main as drv
    with open(os.path.join(ROOT, "jsonfiles/agent/%s.json" % a.agent)) as f:
        sweeps = json.load(f)["sweeps"]
    n_settings = int(np.prod([len(v) for v in sweeps.values()]))
    save = os.path.join(ROOT, "gpurun_out", "%s_%s_device_sweep" % (a.tag, a.agent))
    t0 = time.time()
    data = drv.main(["--env_json", os.path.join(ROOT, "jsonfiles/environment/Pendulum-v0.json"), "--agent_json",
                     os.path.join(ROOT, "jsonfiles/agent/%s.json" % a.agent), "--indices", "0", "1",
                     str(n_settings * a.seeds), "--save_dir", save, "--device_rollout"])
    wall = time.time() - t0
    rows = []
    for hp in sorted(data["experiment_data"]):
        ev = np.stack([np.asarray(r["eval_episode_rewards"]) for r in data["experiment_data"][hp]["runs"]]).mean(axis=-1)
        p = data["experiment_data"][hp]["agent_params"]
        rows.append({"setting": int(hp), "mean_eval_return": float(ev.mean()), "per_seed": [float(x) for x in ev.mean(axis=1)],
                     "last_10_evals": float(ev[:, -10:].mean()), "pi_lr": p["pi_lr"], "qf_vf_lr": p["qf_vf_lr"],
                     "entropy_scale": p["entropy_scale"]})
    rows.sort(key=lambda r: (np.isnan(r["mean_eval_return"]), r["mean_eval_return"]))
    out = {"agent": a.agent, "settings": n_settings, "seeds": a.seeds, "runs": n_settings * a.seeds,
           "steps_per_run": 100000, "wall_s": wall, "reference_tail_plots_ipynb_90": REFERENCE_TAIL[a.agent],
           "table_ascending": rows}
    path = os.path.join(ROOT, "gpurun_out", "%s_%s_device_sweep.json" % (a.tag, a.agent))
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wall %.1f s for %d runs" % (wall, out["runs"]))
    for r in rows[-8:]:
        print("%3d  %9.2f  (pi_lr %g, qf_vf_lr %g, entropy_scale %g)  seeds: %s" %
              (r["setting"], r["mean_eval_return"], r["pi_lr"], r["qf_vf_lr"], r["entropy_scale"],
               " ".join("%.0f" % x for x in r["per_seed"])))
    print("reference tail:", REFERENCE_TAIL[a.agent])


if __name__ == "__main__":
    main()
