#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc pass of SQ counters of bench.py into profiles/<tag>_sq_counters.json.
    python scripts/sq_summary.py --csv <counter_collection.csv> [--csv <second pass>] --updates-per-launch 32 \
           --agents 256 --kernel rlc_ddpg_update_mfma_kernel --kernel-us-per-update 300 --tag r02_mfma
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES
counts cycles summed over the 4 SIMDs of a CU; SQ_INSTS_* count wave-instructions."""
import argparse
import csv
import json
import os
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--csv", action="append", required=True)
    ap.add_argument("--updates-per-launch", type=int, required=True)
    ap.add_argument("--agents", type=int, default=256)
    ap.add_argument("--kernel", default="rlc_ddpg_update_mfma_kernel")
    ap.add_argument("--kernel-us-per-update", type=float, default=None, help="per-CU time of one update (kernel stats)")
    ap.add_argument("--clock-ghz", type=float, default=2.2)
    ap.add_argument("--tag", required=True)
    a = ap.parse_args()
    sums, counts = defaultdict(float), defaultdict(int)
    for path in a.csv:
        with open(path) as f:
            for row in csv.DictReader(f):
                if a.kernel in row.get("Kernel_Name", ""):
                    sums[row["Counter_Name"]] += float(row["Counter_Value"])
                    counts[row["Counter_Name"]] += 1
    mean = {k: sums[k] / counts[k] for k in sums}
    per = float(a.updates_per_launch * a.agents)      # one workgroup (= one CU) per agent
    out = {"kernel": a.kernel, "agents": a.agents, "updates_per_launch": a.updates_per_launch,
           "launches_averaged": {k: counts[k] for k in counts}, "counters_mean_per_launch": mean,
           "per_update_per_cu": {k: mean[k] / per for k in mean}}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in mean:
        busy = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / per / 4.0
        out["mfma_busy_cycles_per_simd_per_update"] = busy
        if "SQ_BUSY_CU_CYCLES" in mean:          # same (profiled) run: no clock assumption
            out["mfma_busy_fraction_of_busy_cu_cycles"] = busy / (mean["SQ_BUSY_CU_CYCLES"] / per)
        if a.kernel_us_per_update:
            out["mfma_busy_fraction"] = busy / (a.kernel_us_per_update * a.clock_ghz * 1e3)
            out["mfma_busy_fraction_note"] = "busy cycles per SIMD / (%.1f us per update x %.2f GHz)" % (
                a.kernel_us_per_update, a.clock_ghz)
    if "SQ_WAVE_CYCLES" in mean:
        wc = mean["SQ_WAVE_CYCLES"]
        out["shares_of_wave_cycles"] = {k: mean[k] / wc for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")
                                        if k in mean}
    path = os.path.join(ROOT, "profiles", "%s_sq_counters.json" % a.tag)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
    print("wrote", path)


if __name__ == "__main__":
    main()
