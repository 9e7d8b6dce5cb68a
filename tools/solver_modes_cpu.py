#!/usr/bin/env python3
"""CPU-only comparison of the oracle's solver back ends on whole alignments, for the reference's
base-station and robot parameter blocks (laser_parameters.yaml:153-301 / 5-147):
  bfgs       the reference's solver on the reference's float objective (gicp.hpp:217-402)
  gn         Gauss-Newton on the per-iteration quadratic form
  bfgs_quad  the reference's BFGS driven by the same quadratic form (what the device's BFGS back
             end runs)
Prints the pose-gap distributions of gn and bfgs_quad against bfgs.
Usage: python tools/solver_modes_cpu.py [config] [pairs] [points]"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from lamp_amd.workload import Workload  # noqa: E402

BLOCKS = {
    "base": dict(corr_dist=0.5, max_iterations=200),
    "robot": dict(corr_dist=0.1, max_iterations=5),
}


def dist(v):
    v = np.asarray(v)
    return "p50 %.1e p90 %.1e max %.1e" % (np.median(v), np.percentile(v, 90), v.max())


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "C2"
    pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    points = int(sys.argv[3]) if len(sys.argv) > 3 else None
    work = Workload(config, pairs=pairs, points=points, keyframes=16)
    for block, kw in BLOCKS.items():
        modes = {
            "bfgs": O.base_gicp_params(solver=O.SOLVER_BFGS, **kw),
            "gn": O.base_gicp_params(solver=O.SOLVER_GN, **kw),
            "bfgs_quad": O.base_gicp_params(solver=O.SOLVER_BFGS, residual_double=1, **kw),
        }

        def one(i):
            a, b = work.pairs[i]
            src, sn = work.keyframes[a]
            tgt, tn = work.keyframes[b]
            return {k: O.gicp_align(p, src, sn, tgt, tn, np.eye(4)) for k, p in modes.items()}

        with ThreadPoolExecutor(max_workers=8) as ex:
            out = list(ex.map(one, range(len(work.pairs))))
        print(f"--- {config} {block}: outer iterations " +
              ", ".join(f"{k} {np.mean([o[k]['iterations'] for o in out]):.1f}" for k in modes))
        for k in ("gn", "bfgs_quad"):
            g = [O.pose_gap(o["bfgs"]["T"], o[k]["T"]) for o in out]
            print(f"{k:10s} vs bfgs: rot {dist([x[0] for x in g])} | trans {dist([x[1] for x in g])}")
        g = [O.pose_gap(o["gn"]["T"], o["bfgs_quad"]["T"]) for o in out]
        print(f"bfgs_quad  vs gn  : rot {dist([x[0] for x in g])} | trans {dist([x[1] for x in g])}")


if __name__ == "__main__":
    main()
