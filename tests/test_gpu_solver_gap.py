"""GPU (Gauss-Newton) against the reference's solver (BFGS, oracle/lo_gicp.c estimate_bfgs) on
whole alignments, per BASELINE configuration.

The reference ends an alignment when pcl::BFGS stops making progress inside the rounding noise
of its float objective (gicp.hpp:307-310, 263-278): its answer is defined only up to that noise.
measure() records, for the same pairs,
  * the gap between the GPU's pose and the BFGS oracle's pose,
  * how far the BFGS oracle's own answer moves when the initial guess moves by one micrometre
    (its self-gap), and the same for the GPU,
  * the gradient norm of the reference's objective at both answers (stationarity),
  * how many pairs leave the GPU's outer loop through its limit-cycle exit and what that exit
    changes.
tools/solver_gap.py runs it for 32 pairs of every configuration and writes
tests/golden/solver_gap.json; the tests below re-measure a subset on the GPU and assert the
bounds that file supports.  Angles use oracle.pose_gap (sin and cos of the angle together).
"""
import json
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "solver_gap.json")

# pairs, points, distinct keyframes measured per configuration (C1's single pair becomes 32 pairs
# of its 10k-point size; C3 / C5 keep their point counts, 32 pairs each)
CASES = {
    "C1": dict(config="C1", pairs=32, points=10000, keyframes=16),
    "C2": dict(config="C2", pairs=32, points=16384, keyframes=16),
    "C3": dict(config="C3", pairs=32, points=32768, keyframes=16),
    "C5": dict(config="C5", pairs=32, points=65536, keyframes=16),
}


def _dist(v):
    v = np.asarray(v, np.float64)
    return {"max": float(v.max()), "p99": float(np.percentile(v, 99)),
            "p90": float(np.percentile(v, 90)), "p50": float(np.median(v))}


def measure(case, threads=8, self_gap=True):
    from lamp_amd import capi
    from lamp_amd.workload import Workload
    c = CASES[case]
    work = Workload(c["config"], pairs=c["pairs"], points=c["points"], keyframes=c["keyframes"])
    pairs = work.pairs

    def gpu_run(no_cycle_exit=False):
        cfg = capi.default_config(True, init_method=capi.INIT_IDENTITY)
        if no_cycle_exit:
            cfg.reserved[5] = 2
        ctx = capi.Context(cfg)
        try:
            work.register(ctx)
            res = ctx.submit(work.candidates())
            st = ctx.stats()
            return [capi.result_to_dict(r) for r in res], int(st.cycle_exits)
        finally:
            ctx.close()

    gpu, n_cycle = gpu_run()
    gpu_nc, _ = gpu_run(no_cycle_exit=True)

    # the same pairs from a guess moved by one micrometre along x
    moved = np.eye(4, dtype=np.float32)
    moved[0, 3] = 1e-6
    gpu_moved = []
    if self_gap:
        ctx = capi.Context(capi.default_config(True, init_method=capi.INIT_IDENTITY))
        try:
            for a, b in pairs:
                src, sn = work.keyframes[a]
                tgt, tn = work.keyframes[b]
                gpu_moved.append(ctx.align(src, sn, tgt, tn, moved)["T"])
        finally:
            ctx.close()

    P = O.base_gicp_params(solver=O.SOLVER_BFGS, use_kdtree=1)

    def cpu_one(i):
        a, b = pairs[i]
        src, sn = work.keyframes[a]
        tgt, tn = work.keyframes[b]
        bf = O.gicp_align(P, src, sn, tgt, tn, np.eye(4))
        bm = O.gicp_align(P, src, sn, tgt, tn, moved) if self_gap else None
        g_gpu = O.gicp_objective_at(P, src, sn, tgt, tn, gpu[i]["T_icp"])
        g_bf = O.gicp_objective_at(P, src, sn, tgt, tn, bf["T"])
        return bf, bm, g_gpu, g_bf

    with ThreadPoolExecutor(max_workers=threads) as ex:
        cpu = list(ex.map(cpu_one, range(len(pairs))))

    rot, trans, rot_bs, trans_bs, rot_gs, trans_gs = [], [], [], [], [], []
    grad_gpu, grad_bf, f_gpu, f_bf = [], [], [], []
    rot_cyc, trans_cyc, n_cyc_diff = [], [], 0
    for i, (bf, bm, g_gpu, g_bf) in enumerate(cpu):
        assert gpu[i]["status"] == 0 and bf["converged"], (i, gpu[i]["status"])
        a, t = O.pose_gap(bf["T"], gpu[i]["T_icp"])
        rot.append(a), trans.append(t)
        if self_gap:
            a, t = O.pose_gap(bf["T"], bm["T"])
            rot_bs.append(a), trans_bs.append(t)
            a, t = O.pose_gap(gpu[i]["T_icp"], gpu_moved[i])
            rot_gs.append(a), trans_gs.append(t)
        grad_gpu.append(float(np.linalg.norm(g_gpu[1]))), f_gpu.append(g_gpu[0])
        grad_bf.append(float(np.linalg.norm(g_bf[1]))), f_bf.append(g_bf[0])
        if gpu[i]["iterations"] != gpu_nc[i]["iterations"]:
            n_cyc_diff += 1
            a, t = O.pose_gap(gpu[i]["T_icp"], gpu_nc[i]["T_icp"])
            rot_cyc.append(a), trans_cyc.append(t)
    out = {
        "case": case, "workload": c, "seed": work.seed, "pairs": len(pairs),
        "gpu_vs_bfgs": {"rot_rad": _dist(rot), "trans_m": _dist(trans)},
        "stationarity_grad_norm": {"gpu": _dist(grad_gpu), "bfgs": _dist(grad_bf)},
        "objective_rel_diff_gpu_minus_bfgs": _dist([abs(fg - fb) / fb for fg, fb in zip(f_gpu, f_bf)]),
        "outer_iterations_mean": {"gpu": float(np.mean([r["iterations"] for r in gpu])),
                                  "bfgs": float(np.mean([c_[0]["iterations"] for c_ in cpu]))},
        "cycle_exit": {"pairs_leaving_through_it": n_cycle,
                       "pairs_whose_iteration_count_differs_without_it": n_cyc_diff,
                       "pose_change_without_it": ({"rot_rad": _dist(rot_cyc), "trans_m": _dist(trans_cyc)}
                                                  if rot_cyc else None)},
    }
    if self_gap:
        out["bfgs_self_gap_guess_moved_1um"] = {"rot_rad": _dist(rot_bs), "trans_m": _dist(trans_bs)}
        out["gpu_self_gap_guess_moved_1um"] = {"rot_rad": _dist(rot_gs), "trans_m": _dist(trans_gs)}
    return out


def _golden():
    return {c["case"]: c for c in json.load(open(GOLDEN))["cases"]}


@pytest.mark.parametrize("case", ["C1", "C2"])
def test_gpu_vs_reference_solver_within_the_reference_own_noise(case):
    """The bound that holds against the reference's solver.  north_star asks for 1e-4 rad /
    1e-4 m: nine pairs in ten meet the rotation bar and the median pair the translation bar;
    the tails (rotation up to 2e-4 rad, translation up to 6e-4 m over 128 measured pairs) are
    the size of the reference's own self-gap (its answer moves by up to a millimetre when the
    guess moves by a micrometre), so no tighter bound on the maximum can hold against it."""
    g = _golden()[case]
    m = measure(case)
    d = m["gpu_vs_bfgs"]
    assert d["rot_rad"]["p90"] < 1e-4 and d["rot_rad"]["max"] < 5e-4, d
    assert d["trans_m"]["p50"] < 1e-4 and d["trans_m"]["max"] < 1e-3, d
    # the gap is not larger than what the reference does to itself under a 1 um change of guess
    bs = m["bfgs_self_gap_guess_moved_1um"]
    assert d["trans_m"]["p50"] <= 3.0 * max(bs["trans_m"]["p50"], 2e-5), (d, bs)
    assert d["trans_m"]["max"] <= 3.0 * max(bs["trans_m"]["max"], 1e-4), (d, bs)
    assert d["rot_rad"]["max"] <= 3.0 * max(bs["rot_rad"]["max"], 2e-5), (d, bs)
    # while the GPU's own answer is stable under the same change
    gs = m["gpu_self_gap_guess_moved_1um"]
    assert gs["trans_m"]["p50"] < 1e-6 and gs["trans_m"]["max"] < 1e-4, gs
    assert gs["rot_rad"]["p50"] < 1e-7 and gs["rot_rad"]["max"] < 1e-4, gs
    # both minimise the same objective: equal values, and the GPU's answer is the more stationary
    s = m["stationarity_grad_norm"]
    assert s["gpu"]["p50"] < 0.1 * s["bfgs"]["p50"], s
    assert s["gpu"]["max"] < 1e-2, s   # the reference's own gradient tolerance (gicp.hpp:250)
    assert m["objective_rel_diff_gpu_minus_bfgs"]["max"] < 5e-3, m["objective_rel_diff_gpu_minus_bfgs"]
    # the limit-cycle exit moves a pose by far less than the reference's noise
    cyc = m["cycle_exit"]["pose_change_without_it"]
    if cyc is not None:
        assert cyc["trans_m"]["max"] < 2e-5 and cyc["rot_rad"]["max"] < 1e-5, cyc
    # the committed distribution is what this measurement reproduces (same seeds, same code)
    assert d["trans_m"]["max"] <= 2.0 * g["gpu_vs_bfgs"]["trans_m"]["max"] + 1e-5
    assert d["rot_rad"]["max"] <= 2.0 * g["gpu_vs_bfgs"]["rot_rad"]["max"] + 1e-6


def test_committed_gap_distribution_covers_every_configuration():
    g = _golden()
    assert sorted(g) == ["C1", "C2", "C3", "C5"]
    for case, m in g.items():
        assert m["pairs"] >= 32
        assert m["gpu_vs_bfgs"]["rot_rad"]["p90"] < 1e-4, (case, m["gpu_vs_bfgs"])
        assert m["gpu_vs_bfgs"]["rot_rad"]["max"] < 5e-4, (case, m["gpu_vs_bfgs"])
        assert m["gpu_vs_bfgs"]["trans_m"]["p50"] < 1e-4, (case, m["gpu_vs_bfgs"])
        assert m["gpu_vs_bfgs"]["trans_m"]["max"] < 1e-3, (case, m["gpu_vs_bfgs"])
