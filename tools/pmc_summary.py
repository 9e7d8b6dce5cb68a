#!/usr/bin/env python3
"""Folds rocprofv3 --pmc counter_collection CSVs (one profiler pass each) into one JSON summary.

usage: pmc_summary.py OUT.json WORKLOAD "COMMAND TEXT" pass1.csv [pass2.csv ...]

Per kernel and counter: launches, sum, mean per launch.  hbm_bytes_per_launch follows the
gfx950 calibration of the FETCH_SIZE / WRITE_SIZE counters (both reported in KiB): FETCH_SIZE
tallies 128-byte requests at 64 bytes, so it is doubled; WRITE_SIZE is taken as is.
"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)          # drop the argument list
    return name.replace("void ", "").strip()


def main():
    out, workload, command, paths = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for p in paths:
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            if not k.startswith("lgicp::"):
                continue
            a = acc[k][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    kernels = {}
    for k, counters in sorted(acc.items()):
        d = {}
        for c, (n, s) in sorted(counters.items()):
            d[c] = {"launches": n, "sum": s, "mean_per_launch": s / max(n, 1)}
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes_per_launch"] = 1024.0 * (2.0 * d["FETCH_SIZE"]["mean_per_launch"] +
                                                  d["WRITE_SIZE"]["mean_per_launch"])
        kernels[k] = d
    json.dump({"command": command, "workload": workload, "kernels": kernels}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
