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
    # Per dispatch of the 1-NN kernel (one profiler pass holds SQ_INSTS_VALU, SQ_WAVES and
    # GRBM_GUI_ACTIVE of the same dispatch): VALU issue utilisation by pass class.  A launch in which
    # most lanes still sweep the tiles executes well over a thousand VALU instructions per wave; one
    # in which most lanes are answered by their certificates a few hundred or fewer.
    per_dispatch = collections.defaultdict(dict)
    for p in paths:
        for r in csv.DictReader(open(p)):
            if "nn_pruned_kernel" in r["Kernel_Name"] and r["Counter_Name"] in ("SQ_INSTS_VALU", "SQ_WAVES",
                                                                                 "GRBM_GUI_ACTIVE"):
                per_dispatch[(p, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    classes = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for d in per_dispatch.values():
        if len(d) < 3 or d["SQ_WAVES"] <= 0 or d["GRBM_GUI_ACTIVE"] <= 0:
            continue
        per_wave = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
        c = classes["sweep (>= 1000 VALU instructions per wave)" if per_wave >= 1000 else
                    "mixed (300..1000)" if per_wave >= 300 else "mostly certified (< 300)"]
        c[0] += 1
        c[1] += d["SQ_INSTS_VALU"]
        c[2] += d["GRBM_GUI_ACTIVE"]
        c[3] += d["SQ_WAVES"]
    pass_classes = {}
    for name, (n, valu, cyc, waves) in classes.items():
        pass_classes[name] = {"launches": n, "valu_instructions_per_wave": valu / waves,
                              "gpu_cycles_per_launch": cyc / 8.0 / n,
                              "valu_issue_utilisation": 4.0 * valu / (1024.0 * cyc / 8.0)}
    kernels = {}
    for k, counters in sorted(acc.items()):
        d = {}
        for c, (n, s) in sorted(counters.items()):
            d[c] = {"launches": n, "sum": s, "mean_per_launch": s / max(n, 1)}
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes_per_launch"] = 1024.0 * (2.0 * d["FETCH_SIZE"]["mean_per_launch"] +
                                                  d["WRITE_SIZE"]["mean_per_launch"])
        kernels[k] = d
    json.dump({"command": command, "workload": workload, "kernels": kernels,
               "nn_pruned_kernel_by_pass_class": pass_classes}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
