#!/usr/bin/env python3
"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database per run; this turns one into the CSV summaries kept under
profiles/:  --stats  -> <out>_kernel_stats.csv  (the --kernel-trace --stats summary: calls, total / average ns, share)
            --pmc    -> <out>_counter_collection.csv (one row per dispatch x counter, the columns pmc_summary.py and
                        sq_summary.py read)
    python scripts/rocpd_extract.py --db gpurun_out/r02_prof/stats/stats_results.db --stats --out profiles/r02"""
import argparse
import csv
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--db", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--pmc", action="store_true")
    ap.add_argument("--kernel", default="rlc_", help="only kernels whose name contains this")
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    if a.stats:
        rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                          "group by name order by sum(duration) desc").fetchall()
        total = float(sum(r[2] for r in rows)) or 1.0
        with open(a.out + "_kernel_stats.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.4f" % (100.0 * r[2] / total), r[4], r[5]])
        print("wrote", a.out + "_kernel_stats.csv")
    if a.pmc:
        rows = db.execute("select dispatch_id, kernel_name, counter_name, value, grid_size, workgroup_size, lds_block_size, "
                          "vgpr_count, sgpr_count, scratch_size from counters_collection where kernel_name like ? "
                          "order by dispatch_id, counter_name", ("%" + a.kernel + "%",)).fetchall()
        with open(a.out + "_counter_collection.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Grid_Size", "Workgroup_Size",
                        "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Scratch_Size"])
            for r in rows:
                w.writerow([r[0], r[1][:160], r[2], r[3]] + list(r[4:]))
        print("wrote", a.out + "_counter_collection.csv", len(rows), "rows")


if __name__ == "__main__":
    main()
