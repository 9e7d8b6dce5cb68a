#!/usr/bin/env python3
"""GPU BFGS back end against the oracle's BFGS (the reference's solver) on a few pairs, base and
robot parameter blocks; prints gap distributions, how many float matrices are bit-identical, and
the submit time.  Usage: python tools/bfgs_probe.py [config] [pairs] [points]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lamp_amd import capi  # noqa: E402
from lamp_amd.workload import Workload  # noqa: E402
from oracle import oracle as O  # noqa: E402


def dist(v):
    v = np.asarray(v)
    return "p50 %.1e p90 %.1e max %.1e" % (np.median(v), np.percentile(v, 90), v.max())


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "C2"
    pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    points = int(sys.argv[3]) if len(sys.argv) > 3 else None
    work = Workload(config, pairs=pairs, points=points, keyframes=16)
    for block, base in (("base", True), ("robot", False)):
        out = {}
        for name, solver in (("gn", capi.SOLVER_GAUSS_NEWTON), ("bfgs", capi.SOLVER_BFGS)):
            cfg = capi.default_config(base, init_method=capi.INIT_IDENTITY, solver=solver,
                                      accumulate_source=0, transform_thresholding=0,
                                      max_tolerable_fitness=1e9)
            ctx = capi.Context(cfg)
            try:
                work.register(ctx)
                ctx.submit(work.candidates())
                t0 = time.perf_counter()
                res = ctx.submit(work.candidates())
                dt = time.perf_counter() - t0
                out[name] = ([capi.result_to_dict(r) for r in res], dt)
            finally:
                ctx.close()
        P = O.base_gicp_params(solver=O.SOLVER_BFGS, corr_dist=0.5 if base else 0.1,
                               max_iterations=200 if base else 5)

        def one(i):
            a, b = work.pairs[i]
            src, sn = work.keyframes[a]
            tgt, tn = work.keyframes[b]
            return O.gicp_align(P, src, sn, tgt, tn, np.eye(4))

        with ThreadPoolExecutor(16) as ex:
            ref = list(ex.map(one, range(len(work.pairs))))
        # the same solver with its sums taken in the order of a 512-lane parallel sum
        P2 = O.base_gicp_params(solver=O.SOLVER_BFGS, corr_dist=0.5 if base else 0.1,
                                max_iterations=200 if base else 5, sum_order=1)

        def two(i):
            a, b = work.pairs[i]
            src, sn = work.keyframes[a]
            tgt, tn = work.keyframes[b]
            return O.gicp_align(P2, src, sn, tgt, tn, np.eye(4))

        with ThreadPoolExecutor(16) as ex:
            ref2 = list(ex.map(two, range(len(work.pairs))))
        g = [O.pose_gap(r["T"], d["T"]) for r, d in zip(ref, ref2)]
        same = sum(np.array_equal(r["T"], d["T"]) for r, d in zip(ref, ref2))
        print(f"{config} {block} oracle bfgs, parallel sum order vs serial: rot {dist([x[0] for x in g])} | "
              f"trans {dist([x[1] for x in g])} | bit-identical T {same}/{len(ref)}", flush=True)
        res, _ = out["bfgs"]
        g = [O.pose_gap(r["T"], d["T_icp"]) for r, d in zip(ref2, res)]
        same = sum(np.array_equal(r["T"], np.asarray(d["T_icp"], np.float32).reshape(4, 4))
                   for r, d in zip(ref2, res))
        print(f"{config} {block} gpu bfgs vs oracle bfgs in parallel sum order: rot {dist([x[0] for x in g])} | "
              f"trans {dist([x[1] for x in g])} | bit-identical T {same}/{len(res)}", flush=True)
        for name in ("gn", "bfgs"):
            res, dt = out[name]
            g = [O.pose_gap(r["T"], d["T_icp"]) for r, d in zip(ref, res)]
            same = sum(np.array_equal(r["T"], np.asarray(d["T_icp"], np.float32).reshape(4, 4))
                       for r, d in zip(ref, res))
            its = sum(r["iterations"] == d["iterations"] for r, d in zip(ref, res))
            print(f"{config} {block} gpu {name:4s} vs oracle bfgs: rot {dist([x[0] for x in g])} | "
                  f"trans {dist([x[1] for x in g])} | bit-identical T {same}/{len(res)}, "
                  f"equal iteration counts {its}/{len(res)}, submit {1e3 * dt:.1f} ms", flush=True)


if __name__ == "__main__":
    main()
