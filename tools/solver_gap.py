#!/usr/bin/env python3
"""Measures the GPU (Gauss-Newton) against the reference's solver (BFGS oracle) on 32 pairs of
every BASELINE configuration and writes the distributions as JSON (tests/golden/solver_gap.json is
a copy of this script's output from an MI355X box).  Usage: python tools/solver_gap.py OUT.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.test_gpu_solver_gap import CASES, measure  # noqa: E402


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "solver_gap.json")
    threads = min(16, os.cpu_count() or 1)
    cases = []
    for case in CASES:
        t0 = time.time()
        m = measure(case, threads=threads)
        m["seconds"] = round(time.time() - t0, 1)
        cases.append(m)
        print(case, json.dumps(m["gpu_vs_bfgs"]), f"{m['seconds']} s", flush=True)
        json.dump({"generator": "tools/solver_gap.py", "cases": cases}, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
