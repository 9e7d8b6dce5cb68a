#!/usr/bin/env python3
"""Measures both back ends of the GPU (Gauss-Newton, BFGS) against the reference's solver (BFGS
oracle) on 32 pairs of every BASELINE configuration, under the reference's base-station and robot
parameter blocks, and writes the distributions as JSON (tests/golden/solver_gap.json is a copy of
this script's output from an MI355X box).
Usage: python tools/solver_gap.py OUT.json [bfgs]   (bfgs: only the BFGS back end's part; the
Gauss-Newton part is taken over from tests/golden/solver_gap.json)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.test_gpu_solver_gap import BLOCKS, CASES, measure, measure_bfgs  # noqa: E402


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "solver_gap.json")
    threads = min(16, os.cpu_count() or 1)
    cases, bfgs = [], []
    only_bfgs = len(sys.argv) > 2 and sys.argv[2] == "bfgs"
    if only_bfgs:  # keep the Gauss-Newton part of the committed file
        cases = json.load(open(os.path.join(ROOT, "tests", "golden", "solver_gap.json")))["cases"]
    for case in CASES:
        for block in BLOCKS:
            t0 = time.time()
            m = measure_bfgs(case, block, threads=threads)
            m["seconds"] = round(time.time() - t0, 1)
            bfgs.append(m)
            print(case, block, json.dumps({k: m[k] for k in m if k.startswith(("gpu_", "oracle_"))}),
                  f"{m['seconds']} s", flush=True)
            json.dump({"generator": "tools/solver_gap.py", "cases": cases, "bfgs_back_end": bfgs},
                      open(out_path, "w"), indent=1)
    for case in ([] if only_bfgs else CASES):
        t0 = time.time()
        m = measure(case, threads=threads)
        m["seconds"] = round(time.time() - t0, 1)
        cases.append(m)
        print(case, json.dumps(m["gpu_vs_bfgs"]), f"{m['seconds']} s", flush=True)
        json.dump({"generator": "tools/solver_gap.py", "cases": cases, "bfgs_back_end": bfgs},
                  open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
