#!/usr/bin/env python3
"""Merge scripts/micro/peaks' JSON line with its rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<tag>_peaks.json:
measured roofs of the box + the counter calibration factors (counter bytes / bytes the kernel really moved).
    python scripts/peaks_summary.py --plain gpurun_out/r03/peaks.json --fetch <csv> --write <csv> --tag r03"""
import argparse
import csv
import json
import os
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"].split("(")[0].split("<")[0].strip()].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plain", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--tag", default="r03")
    a = ap.parse_args()
    out = json.loads(open(a.plain).read().strip().splitlines()[-1])
    nbytes = float(out["buffer_bytes"])
    out["spec"] = {"fp32_matrix_tflops": 157.3, "hbm_gbs": 8000.0}
    if a.fetch:
        f = per_kernel(a.fetch, "FETCH_SIZE")
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB
        out["fetch_size_over_bytes_read"] = {k: v * 1024.0 / nbytes for k, v in f.items() if "calib" in k or "copy" in k or "read" in k}
    if a.write:
        w = per_kernel(a.write, "WRITE_SIZE")
        out["write_size_over_bytes_written"] = {k: v * 1024.0 / nbytes for k, v in w.items() if "rw" in k or "copy" in k}
    out["note"] = ("calib_kloop_dword = the forward k-loop's weight stream (four dwords 16 B apart per lane per 1 KB block); calib_b128 = "
                   "16 B per lane; factors = counter bytes / bytes actually read (written) once from a buffer larger than the "
                   "Infinity Cache")
    path = os.path.join(ROOT, "profiles", "%s_peaks.json" % a.tag)
    with open(path, "w") as fo:
        json.dump(out, fo, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
