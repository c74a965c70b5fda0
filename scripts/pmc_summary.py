#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into profiles/<tag>_pmc_traffic.json.
    python scripts/pmc_summary.py --fetch <counter_collection.csv> --write <counter_collection.csv> \
           --updates-per-launch 32 --agents 256 --tag r01
Corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are KB at the L2's
memory side (Infinity-Cache hits included); on gfx950 FETCH_SIZE reports half of the bytes of 16 B/lane
streaming reads, WRITE_SIZE is exact for 16 B/lane stores; other widths are uncalibrated.  The fused kernel mixes
16 B/lane reads (Adam state, backward GEMM) with dword reads (forward GEMM weights), so the corrected fetch is
reported as an interval [raw, 2 x raw]; bench.py quotes the upper end."""
import argparse
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mean_kb(path, counter, kernel_substr):
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") == counter and kernel_substr in row.get("Kernel_Name", ""):
                vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit("no %s rows for %s in %s" % (counter, kernel_substr, path))
    return sum(vals) / len(vals), len(vals)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--updates-per-launch", type=int, default=32)
    ap.add_argument("--agents", type=int, default=256)
    ap.add_argument("--kernel", default="rlc_ddpg_update_mfma_kernel")
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--algorithmic-bytes", type=float, default=None, help="per update; default by kernel name")
    a = ap.parse_args()
    algo_bytes = a.algorithmic_bytes
    if algo_bytes is None:      # bench.py's figures
        algo_bytes = {"rlc_sac": 51716 * 32 + 3600.0, "rlc_naf": 83406 * 32 + 8000.0}.get(a.kernel[:7], 2634064.0)
    f_kb, nf = mean_kb(a.fetch, "FETCH_SIZE", a.kernel)
    w_kb, nw = mean_kb(a.write, "WRITE_SIZE", a.kernel)
    per = a.updates_per_launch * a.agents
    out = {
        "kernel": a.kernel, "launches_averaged": [nf, nw], "updates_per_launch": a.updates_per_launch, "agents": a.agents,
        "fetch_size_raw_bytes_per_launch": f_kb * 1024.0, "write_size_bytes_per_launch": w_kb * 1024.0,
        "fetch_raw_bytes_per_update": f_kb * 1024.0 / per, "fetch_corrected_upper_bytes_per_update": 2 * f_kb * 1024.0 / per,
        "write_bytes_per_update": w_kb * 1024.0 / per,
        "traffic_bytes_per_launch_upper": (2 * f_kb + w_kb) * 1024.0,
        "traffic_bytes_per_launch_lower": (f_kb + w_kb) * 1024.0,
        "algorithmic_bytes_per_update": algo_bytes,
        "traffic_bytes_per_update": (2 * f_kb + w_kb) * 1024.0 / per,
        "traffic_over_algorithmic": (2 * f_kb + w_kb) * 1024.0 / per / algo_bytes,
        "note": "traffic = 2 x FETCH_SIZE + WRITE_SIZE: profiles/r03_peaks.json calibrates FETCH_SIZE at one half of the bytes read "
                "for the k-loop's dword stream as well as for 16 B/lane streams (the *_lower field is kept for older readers)",
    }
    path = os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % a.tag)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
    print("wrote", path)


if __name__ == "__main__":
    main()
