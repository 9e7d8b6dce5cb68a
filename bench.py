#!/usr/bin/env python3
"""bench.py — loop-closure GICP pairs verified per second on MI355X.

One step = one lgicp_submit() of a whole candidate batch (every GICP outer iteration to
convergence, fitness, correspondences, Ap / information) with the batch's keyframes already
resident in HBM.  N=1 runs BASELINE.json configs[1] (C2: 256 candidate pairs, 16k-point
VLP-16-like keyframes).  The timed steps rotate through several distinct batches (default 4:
256 keyframes, a working set beyond the 256 MiB Infinity Cache), so a step never meets the
data of the step before it in cache.

What the JSON line adds to the contract's fields:
  value_single_submit           pairs/s with ONE lgicp_submit in flight: what a LAMP node gets, whose
                                ComputeTransforms blocks once per tick (`value` keeps several in flight)
  accumulated / solver_bfgs /   sub-records of the default C2 line (own contexts, one submit at a
  c3_list                       time): consecutive keys so that targets accumulate key +- 2; the BFGS
                                back end; north_star's C3 list (1,000 pairs x 32k points) on this GPU
                                or, with --gpus N, cut over the N ranks with the RCCL gather
  registration_ms_per_keyframe  upload + spatial index + covariances of one keyframe (batched
                                lgicp_add_keyframes, measured before the timed region) — the work
                                the reference repeats inside every alignment
  value_cold                    pairs/s of a batch whose keyframes have never been seen:
                                pairs / (registration of its keyframes + its submit)
  roofline                      dominant kernel (1-NN lookups), from a phase that runs the same
                                batches on ONE stream, so kernel durations are exclusive
  roofline_reduce               the HBM-bound linearise/reduce kernel, same phase
  cpu_baseline                  the CPU oracle on the same pairs, in the reference's
                                configuration and in a non-oversubscribed one

N>1: --mode weak (default) gives every rank its own batches of the same shape; --mode strong
cuts ONE candidate list (e.g. --config C3: 1,000 pairs over 250 keyframes of 32k points) over the
ranks with lamp_amd.sharding.partition_pairs, every rank registers only the keyframes its pairs
touch, and rank 0 gathers the variable-length result records over RCCL.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFLOPS = 157.3   # MI355X vector FP32 peak (packed FMA), per GPU
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E, per GPU
KEY_STRIDE = 1000000            # key offset between the batches living in one context


def pmc_summary(config):
    """The newest committed rocprofv3 --pmc summary of this workload (profiles/*_pmc_summary.json;
    FETCH_SIZE and WRITE_SIZE are collected in their own profiler passes, not in this process)."""
    d = os.path.join(ROOT, "profiles")
    best = None
    for name in sorted(os.listdir(d)) if os.path.isdir(d) else []:
        if name.endswith("_pmc_summary.json") and f"_{config.lower()}_" in name:
            best = os.path.join(d, name)
    if best is None:
        return None, None
    try:
        return json.load(open(best)), os.path.basename(best)
    except Exception:
        return None, None


def pmc_kernel(summary, kernel):
    """Counters of one kernel; the instantiations of a kernel template (name<...>) are folded into
    one entry, their per-launch figures weighted by their launches."""
    if not summary:
        return {}
    ks = summary.get("kernels", {})
    if kernel in ks:
        return ks[kernel]
    parts = [v for k, v in ks.items() if k.startswith(kernel + "<")]
    if not parts:
        return {}
    out = {}
    for c in set().union(*[set(p) for p in parts]):
        rows = [p[c] for p in parts if isinstance(p.get(c), dict)]
        if rows:
            n = sum(r["launches"] for r in rows)
            tot = sum(r["sum"] for r in rows)
            out[c] = {"launches": n, "sum": tot, "mean_per_launch": tot / max(n, 1)}
    if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
        out["hbm_bytes_per_launch"] = 1024.0 * (2.0 * out["FETCH_SIZE"]["mean_per_launch"] +
                                                out["WRITE_SIZE"]["mean_per_launch"])
    return out


def cpu_threads_available():
    """Host threads this job may use: the CPU affinity, capped by the box's own thread limit
    (OMP_NUM_THREADS is set to the job's CPU share on the GPU box; os.cpu_count() shows the whole
    machine there)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    cap = os.environ.get("OMP_NUM_THREADS")
    if cap and cap.isdigit() and int(cap) > 0:
        n = min(n, int(cap))
    return max(1, n)


def cpu_baseline_c4(work, n_sample, cores, harris_radius, method="teaser"):
    """C4 on the CPU: the oracle's feature front end (Harris keypoints, FPFH, descriptor matches,
    TEASER++ restatement or SAC-IA) from the odometry guess, then GICP (kd-tree, BFGS) from its
    result — one pair per host thread."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.lib()
    pairs = work.pairs[:n_sample]

    def pose(k):
        return O.pose_from_quat(*work.poses[k])

    def one(pair):
        a, b = pair
        src, sn = work.keyframes[a]
        tgt, tn = work.keyframes[b]
        p21 = O.pose_between(pose(b), pose(a))
        g = np.eye(4, dtype=np.float32)
        g[:3, :3] = np.array(p21.R).reshape(3, 3)
        g[:3, 3] = np.array(p21.t)
        if method == "sac":
            T0, fit0, _ = O.sac_initial_alignment(src, sn, tgt, tn, harris_radius=harris_radius,
                                                  iterations=500, guess=g, seed=a * 7919 + b)
            if not fit0 < 32.0:      # sac_fitness_score_threshold of the base block: rejected
                return False
        else:
            T0, used, nm, nc = O.teaser_initial_alignment(src, sn, tgt, tn, harris_radius=harris_radius,
                                                          guess=g)
        r = O.gicp_align(O.base_gicp_params(solver=O.SOLVER_BFGS, use_kdtree=1), src, sn, tgt, tn,
                         guess=T0)
        return bool(r["converged"])

    t0 = time.perf_counter()
    st = []
    with ThreadPoolExecutor(max_workers=min(cores, len(pairs))) as ex:
        for k, ok in enumerate(ex.map(one, pairs)):
            st.append(ok)
            print(f"[bench] cpu baseline pair {k + 1}/{len(pairs)} after {time.perf_counter() - t0:.1f} s",
                  file=sys.stderr, flush=True)
    dt = time.perf_counter() - t0
    return {"value": len(pairs) / dt, "unit": "pairs/s", "cores": min(cores, len(pairs)), "kind": "port",
            "host_cpus_visible": os.cpu_count(),
            "sample": f"first {len(pairs)} pairs of the same workload, one pair per host thread: CPU "
                      f"oracle front end (Harris, FPFH, {'SAC-IA' if method == 'sac' else 'TEASER++ restatement'}) "
                      f"+ GICP (kd-tree, BFGS); "
                      f"{sum(st)} converged"}


def cpu_baseline(work, n_sample, cores):
    """The CPU oracle (exact kd-tree lookups, BFGS: the reference's algorithm) on the first n_sample
    pairs of the same workload, timed like eval_loop_computation_test.cc:338-344, in two
    configurations: the reference's base station (pool of 0.8 x cores pair workers, 8 OpenMP
    threads per alignment: laser_parameters.yaml:222,238) and a non-oversubscribed one (one pair
    per core, 1 OpenMP thread)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.lib()
    pairs = work.pairs[:n_sample]
    ident = O.pose_from_quat([1, 0, 0, 0], [0, 0, 0])

    def run(pool, omp):
        params = O.base_icp_params(init_method=O.INIT_IDENTITY, solver=O.SOLVER_BFGS, threads=omp,
                                   transform_thresholding=0)

        def one(pair):
            a, b = pair
            src, sn = work.keyframes[a]
            tgt, tn = work.keyframes[b]
            return O.perform_alignment(params, src, sn, tgt, tn, src, tn, ident)["status"]

        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=pool) as ex:
            st = list(ex.map(one, pairs))
        dt = time.perf_counter() - t0
        return len(pairs) / dt, sum(1 for s in st if s == 0)

    ref_pool = max(1, int(0.8 * cores))
    v_ref, ok_ref = run(ref_pool, 8)
    v_flat, ok_flat = run(cores, 1)
    best = max(v_ref, v_flat)
    return {
        "value": best, "unit": "pairs/s", "cores": cores, "kind": "port",
        "host_cpus_visible": os.cpu_count(),
        "reference_configuration": {"pool_threads": ref_pool, "omp_threads_per_alignment": 8,
                                    "value": v_ref, "verified": ok_ref},
        "one_pair_per_core": {"pool_threads": cores, "omp_threads_per_alignment": 1,
                              "value": v_flat, "verified": ok_flat},
        "sample": f"first {len(pairs)} pairs of the same workload; CPU oracle (kd-tree lookups and "
                  f"kd-tree build per alignment, BFGS), value = the faster of the two configurations",
    }


def run_steps_in_flight(n_steps, first, depth, submit_step, on_result):
    """Steps first .. first + n_steps - 1 with `depth` of them in flight: lane k (a thread of its
    own) runs submit_step(k, i) for i = first + k, first + k + depth, ... without waiting for the
    other lanes; on_result(i, value) is called on the calling thread in step order (with several
    ranks the gather behind it is a collective: every rank must issue them in the same order).  An
    exception in a lane ends the run and is raised here."""
    import threading
    done, cv, failed = {}, threading.Condition(), []

    def lane_loop(k):
        try:
            for i in range(first + k, first + n_steps, depth):
                value = submit_step(k, i)
                with cv:
                    if failed:
                        return
                    done[i] = value
                    cv.notify_all()
        except Exception as e:   # noqa: BLE001 - raised again by the calling thread
            with cv:
                failed.append(e)
                cv.notify_all()

    threads = [threading.Thread(target=lane_loop, args=(k,)) for k in range(depth)]
    for t in threads:
        t.start()
    try:
        for i in range(first, first + n_steps):
            with cv:
                cv.wait_for(lambda: i in done or failed)
                if failed:
                    break
                value = done.pop(i)
            on_result(i, value)
    except Exception as e:       # noqa: BLE001 - the lanes are told to stop, then it is raised
        with cv:
            failed.append(e)
    finally:
        for t in threads:
            t.join()
    if failed:
        raise failed[0]


def make_config(capi, args, local_rank, slots=0, stage_timing=False):
    if args.config == "C4":
        cfg = capi.default_config(True, device=local_rank,
                                  init_method=capi.INIT_FEATURES if args.c4_init == "sac" else capi.INIT_TEASERPP,
                                  harris_radius=args.harris_radius, transform_thresholding=0)
    else:
        cfg = capi.default_config(True, device=local_rank, init_method=capi.INIT_IDENTITY)
    if args.solver == "bfgs":
        cfg.solver = capi.SOLVER_BFGS
    if args.nn_variant:
        cfg.reserved[0] = args.nn_variant
    if args.exhaustive:
        cfg.reserved[2] = 1
    if slots:
        cfg.reserved[7] = slots
    # per-stage hipEvent times (lgicp_stats nn_ms ...): the events sit between the kernels of every
    # outer iteration and cost a lone C2 submit 5-7 %, so the timed region runs without them unless
    # no exclusive phase will measure the stages
    cfg.stage_timing = 1 if stage_timing else 0
    return cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=80)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--mode", default="weak", choices=["weak", "strong"])
    ap.add_argument("--batches", type=int, default=0,
                    help="distinct batches the steps rotate through (0 = 4 for C2, 1 otherwise)")
    ap.add_argument("--pairs", type=int, default=None)
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--keyframes", type=int, default=None)
    ap.add_argument("--harris-radius", type=float, default=0.5, help="C4 only")
    ap.add_argument("--c4-init", default="teaser", choices=["teaser", "sac"],
                    help="C4 only: TEASER++ (the configuration's method) or SAC-IA global initialisation")
    ap.add_argument("--cpu-sample", type=int, default=256,
                    help="pairs of the same workload timed on the host CPU")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-exclusive-phase", action="store_true",
                    help="skip the one-stream phase the roofline figures come from")
    ap.add_argument("--no-cold", action="store_true")
    ap.add_argument("--nn-variant", type=int, default=0)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and run the result gather even with one rank")
    ap.add_argument("--slots", type=int, default=0,
                    help="parts of a submit in flight at once (0 = the library's default)")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="submits in flight at once, each from its own host thread "
                         "(lgicp_set_submit_lanes; 0 = 6 for the GICP-only configurations, 2 for C4)")
    ap.add_argument("--no-sub-records", action="store_true",
                    help="skip the extra workloads of the default C2 line (accumulated targets, the "
                         "whole C3 list, the BFGS back end)")
    ap.add_argument("--solver", default="gn", choices=["gn", "bfgs"],
                    help="inner solver of the headline workload (lgicp_config.solver)")
    ap.add_argument("--exhaustive", action="store_true",
                    help="use the exhaustive 1-NN kernel instead of the pruned one")
    args = ap.parse_args()

    t_start = time.perf_counter()
    # stdout carries the one JSON line and nothing else: libraries that print banners there (RCCL
    # does at communicator creation) are pointed at stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def progress(what):
        # a line on stderr per phase: long configurations (C4, C5) stay visibly alive
        if int(os.environ.get("RANK", "0")) == 0:
            print(f"[bench {time.perf_counter() - t_start:7.1f} s] {what}", file=sys.stderr, flush=True)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or args.force_dist:
        # torch first: liblamp_gicp.so then binds to the HIP runtime torch has already loaded
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from lamp_amd import capi
    from lamp_amd.sharding import RecordGatherer, pack_results, shard_workload
    from lamp_amd.workload import Workload

    strong = args.mode == "strong"
    # (C4 keeps Harris keypoints and FPFH descriptors per keyframe: two batches in rotation, so that
    # a step does not simply meet the features its predecessor left; value_cold has none cached)
    n_batches = args.batches or ((4 if args.config == "C2" else (2 if args.config == "C4" else 1))
                                 if not strong else 1)
    # ---- workloads: the batches this rank owns ------------------------------------------------
    works, cand_lists, key_lists = [], [], []
    for b in range(n_batches):
        # weak: every rank its own data; strong: the same list on every rank, cut below
        shard = b if strong else rank * 64 + b
        w = Workload(args.config, pairs=args.pairs, points=args.points, keyframes=args.keyframes,
                     shard=shard, key_offset=b * KEY_STRIDE, progress=progress)
        if strong:
            mine, keys = shard_workload(w, world, rank)
        else:
            mine, keys = list(range(len(w.pairs))), sorted(w.keyframes)
        works.append(w)
        cand_lists.append([w.pairs[i] for i in mine])
        key_lists.append(keys)
    total_pairs_per_batch = len(works[0].pairs) if strong else len(works[0].pairs) * world

    main_stage_timing = bool(args.no_exclusive_phase or args.exhaustive or args.config == "C4")
    ctx = capi.Context(make_config(capi, args, local_rank, args.slots, stage_timing=main_stage_timing))
    # Steps overlap: the last outer iterations of a batch (a handful of pairs, latency-bound) run
    # beside the full-size launches of the next ones.  A context with a feature-based initialisation
    # (C4) takes two submits at a time.
    in_flight = args.in_flight or (1 if args.slots == 1 else (2 if args.config == "C4" else 6))

    def register(c, w, keys):
        t0 = time.perf_counter()
        c.add_keyframes(keys, [w.keyframes[k][0] for k in keys], [w.keyframes[k][1] for k in keys])
        for k in keys:
            c.add_pose(k, *w.poses[k])
        c.wait_registered()
        return time.perf_counter() - t0

    reg_s = reg_kf = 0
    for w, keys in zip(works, key_lists):     # keyframes + poses resident in HBM before timing
        reg_s += register(ctx, w, keys)
        reg_kf += len(keys)
    packed = [ctx.pack_candidates(c) for c in cand_lists]
    progress(f"{reg_kf} keyframes registered in {reg_s:.3f} s")

    gatherer = None
    if dist is not None:
        import torch
        counts = [len(cand_lists[0])] * world
        if strong:
            t = torch.zeros(world, dtype=torch.int64, device="cuda")
            t[rank] = len(cand_lists[0])
            dist.all_reduce(t)
            counts = [int(v) for v in t.cpu()]
        gatherer = RecordGatherer(dist, rank, world, counts, torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    def gather(i, out):
        if dist is not None:
            import torch
            gatherer.gather(pack_results(out, len(cand_lists[i % n_batches])))   # rank 0 ends with every record
            torch.cuda.synchronize()

    def step(i, c=None, pk=None):
        b = i % n_batches
        out = (c or ctx).submit_packed((pk or packed)[b])   # blocking: returns after the device work
        if c is None:
            gather(i, out)
        return out

    def collect_overlapped(n_steps, first, depth):
        """`depth` steps in flight, each lane of steps submitted by its own thread; results are
        taken (and, with several ranks, gathered) in step order by this thread.  Stage times are
        those of the step's own submit (lgicp_submit_stats); the launch and distance counters are
        exact differences over the region."""
        acc = dict(nn_ms=0.0, nn_launches=0.0, nn_evals=0.0, nn_exec=0.0, red_ms=0.0,
                   red_launches=0.0, red_points=0.0, red_inliers=0.0, cycle_exits=0.0,
                   clique_ms=0.0, clique_bytes=0.0, clique_launches=0.0,
                   core_ms=0.0, core_bytes=0.0, core_launches=0.0)
        stage = dict(nn=0.0, reduce=0.0, solve=0.0, cov=0.0, post=0.0, total=0.0)
        s_begin = ctx.stats()
        res = []

        def submit_step(k, i):
            out, st = ctx.submit_packed_stats(packed_sets[k][i % n_batches])
            rec = pack_results(out, len(cand_lists[i % n_batches])) if dist is not None else None
            return out, rec, st

        def on_result(i, value):
            out, rec, s1 = value
            if rec is not None:
                import torch
                gatherer.gather(rec)
                # the collective's own stream only: a device-wide wait would also sit out whatever
                # the other lanes have in flight (the closing barrier() waits for the device)
                torch.cuda.current_stream().synchronize()
            res.append(out)
            acc["nn_ms"] += s1.nn_ms
            acc["nn_exec"] += s1.nn_executed_evals
            acc["red_ms"] += s1.reduce_ms
            acc["red_launches"] += s1.reduce_launches
            acc["red_points"] += s1.reduce_points
            acc["red_inliers"] += s1.reduce_inliers
            acc["cycle_exits"] += s1.cycle_exits
            for k in stage:
                stage[k] += getattr(s1, k + "_ms")

        run_steps_in_flight(n_steps, first, depth, submit_step, on_result)
        s_end = ctx.stats()
        acc["nn_launches"] = s_end.nn_launches - s_begin.nn_launches
        acc["nn_evals"] = s_end.nn_distance_evals - s_begin.nn_distance_evals
        return acc, stage, res

    def collect(c, n_steps, first, pk=None):
        acc = dict(nn_ms=0.0, nn_launches=0.0, nn_evals=0.0, nn_exec=0.0, red_ms=0.0,
                   red_launches=0.0, red_points=0.0, red_inliers=0.0, cycle_exits=0.0,
                   clique_ms=0.0, clique_bytes=0.0, clique_launches=0.0,
                   core_ms=0.0, core_bytes=0.0, core_launches=0.0)
        stage = dict(nn=0.0, reduce=0.0, solve=0.0, cov=0.0, post=0.0, total=0.0)
        res = []
        for i in range(first, first + n_steps):
            s0 = c.stats()
            t_s = time.perf_counter()
            res.append(step(i, None if c is ctx else c, pk))
            if time.perf_counter() - t_s > 1.0:
                progress(f"step {i - first + 1}/{n_steps}: {time.perf_counter() - t_s:.1f} s")
            s1 = c.stats()
            acc["nn_ms"] += s1.nn_ms
            acc["nn_launches"] += s1.nn_launches - s0.nn_launches
            acc["nn_evals"] += s1.nn_distance_evals - s0.nn_distance_evals
            acc["nn_exec"] += s1.nn_executed_evals
            acc["red_ms"] += s1.reduce_ms
            acc["red_launches"] += s1.reduce_launches
            acc["red_points"] += s1.reduce_points
            acc["red_inliers"] += s1.reduce_inliers
            acc["cycle_exits"] += s1.cycle_exits
            acc["clique_ms"] += s1.clique_ms
            acc["clique_bytes"] += s1.clique_row_bytes
            acc["clique_launches"] += s1.clique_launches
            acc["core_ms"] += s1.core_ms
            acc["core_bytes"] += s1.core_matrix_bytes
            acc["core_launches"] += s1.core_launches
            for k in stage:
                stage[k] += getattr(s1, k + "_ms")
        return acc, stage, res

    for i in range(args.warmup):
        step(i)
        progress(f"warm-up {i + 1}/{args.warmup}")
    serial = ser_acc = None
    if in_flight > 1:
        # What a LAMP node gets: one blocking submit per tick on a context as lgicp_create leaves
        # it (one lane).  Measured first; the lanes for the steps in flight are set up afterwards.
        n_ser = max(2 * n_batches, min(args.steps, 16))
        barrier()
        t_ser = time.perf_counter()
        ser_acc, _, _ = collect(ctx, n_ser, 0)
        ser_acc["steps"] = n_ser
        barrier()
        serial = (time.perf_counter() - t_ser) / n_ser
        in_flight = ctx.set_submit_lanes(in_flight)
    # one set of candidate / result arrays per submit in flight: step i uses set i mod in_flight
    packed_sets = [packed] + [[ctx.pack_candidates(c) for c in cand_lists] for _ in range(in_flight - 1)]
    if in_flight > 1:
        # every set of arrays and every lane's buffers sized before the timed region
        collect_overlapped(2 * in_flight, 0, in_flight)
    barrier()
    t0 = time.perf_counter()
    if in_flight > 1:
        acc, stage, results = collect_overlapped(args.steps, args.warmup, in_flight)
    else:
        acc, stage, results = collect(ctx, args.steps, args.warmup)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # accuracy of the last visit of every batch against the known offsets (sanity, not parity)
    ok = n_res = 0
    iters, terr, rerr = [], [], []
    from lamp_amd.workload import pose_gap
    seen = {}
    for i, res in enumerate(results):
        seen[(args.warmup + i) % n_batches] = res
    for b, res in seen.items():
        for r, (a, bb) in zip([res[k] for k in range(len(cand_lists[b]))], cand_lists[b]):
            n_res += 1
            iters.append(r.iterations)
            if r.status == 0:
                ok += 1
                ang, te = pose_gap(works[b].truth[(a, bb)], np.array(r.T_icp, np.float64).reshape(4, 4))
                terr.append(te), rerr.append(ang)

    progress(f"timed region: {dt:.3f} s for {args.steps} steps")
    # ---- a batch never seen before: registration + submit ----------------------------------------
    cold = None
    if not args.no_cold and not strong and world == 1:
        wc = Workload(args.config, pairs=args.pairs, points=args.points, keyframes=args.keyframes,
                      shard=rank * 64 + n_batches, key_offset=n_batches * KEY_STRIDE)
        keys_c = sorted(wc.keyframes)
        pk_c = ctx.pack_candidates(list(wc.pairs))
        t_a = time.perf_counter()
        register(ctx, wc, keys_c)
        t_b = time.perf_counter()
        ctx.submit_packed(pk_c)
        t_c = time.perf_counter()
        cold = {"keyframes": len(keys_c), "pairs": len(wc.pairs), "registration_ms": 1e3 * (t_b - t_a),
                "submit_ms": 1e3 * (t_c - t_b), "value": len(wc.pairs) / (t_c - t_a)}

    progress("cold batch done" if cold else "no cold batch")
    # ---- exclusive phase: the same batches on one stream -----------------------------------------
    excl = None
    # (C4: the TEASER++ stage, where its time goes, runs on one stream with only the next group's graph kernels beside it)
    if not args.no_exclusive_phase and rank == 0 and not args.exhaustive and args.config != "C4":
        c1 = capi.Context(make_config(capi, args, local_rank, slots=1, stage_timing=True))
        try:
            for w, keys in zip(works, key_lists):
                register(c1, w, keys)
            pk1 = [c1.pack_candidates(c) for c in cand_lists]
            n_excl = max(n_batches, min(args.steps, 2 * n_batches))
            for i in range(n_batches):
                c1.submit_packed(pk1[i])
            t1 = time.perf_counter()
            e_acc, e_stage, _ = collect(c1, n_excl, 0, pk1)
            e_dt = time.perf_counter() - t1
            excl = dict(acc=e_acc, stage=e_stage, steps=n_excl, ms_per_step=1e3 * e_dt / n_excl)
        finally:
            c1.close()

    progress("exclusive phase done" if excl else "no exclusive phase")

    # ---- sub-records: other workloads through contexts of their own, one submit at a time ----------
    def sub_record(config, what, *, pairs=None, points=None, keyframes=None, consecutive_keys=False,
                   solver=None, cut_over_ranks=False, steps=6, warm=2, exclusive_nn=False):
        """One more workload, submitted the way a LAMP node submits: one blocking lgicp_submit per
        tick (IcpLoopComputation::ComputeTransforms, IcpLoopComputation.cc:243-367).  With
        cut_over_ranks every rank takes its share of ONE candidate list and rank 0 gathers the
        result records over RCCL inside the timed region; otherwise only rank 0 runs it."""
        if not cut_over_ranks and rank != 0:
            return None
        w = Workload(config, pairs=pairs, points=points, keyframes=keyframes, shard=977,
                     consecutive_keys=consecutive_keys, progress=progress)
        if cut_over_ranks:
            mine, keys = shard_workload(w, world, rank)
        else:
            mine, keys = list(range(len(w.pairs))), sorted(w.keyframes)
        cands = [w.pairs[i] for i in mine]
        cfg = capi.default_config(True, device=local_rank, init_method=capi.INIT_IDENTITY)
        if solver is not None:
            cfg.solver = solver
        c = capi.Context(cfg)
        try:
            t_reg = register(c, w, keys)
            pk = c.pack_candidates(cands)
            g = None
            if cut_over_ranks and dist is not None:
                import torch
                t = torch.zeros(world, dtype=torch.int64, device="cuda")
                t[rank] = len(cands)
                dist.all_reduce(t)
                g = RecordGatherer(dist, rank, world, [int(v) for v in t.cpu()],
                                   torch.device("cuda", local_rank))

            def one():
                out = c.submit_packed(pk)
                if g is not None:
                    import torch
                    g.gather(pack_results(out, len(cands)))
                    torch.cuda.synchronize()
                return out

            for _ in range(warm):
                one()
            if cut_over_ranks:
                barrier()
            t_a = time.perf_counter()
            for _ in range(steps):
                out = one()
            if cut_over_ranks:
                barrier()
            d = time.perf_counter() - t_a
            if cut_over_ranks and dist is not None:
                import torch
                tt = torch.tensor([d], dtype=torch.float64, device="cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                d = float(tt.item())
            st = c.stats()
            n_ok = sum(1 for k in range(len(cands)) if out[k].status == 0)
            its = [out[k].iterations for k in range(len(cands))]
            tgt_pts = float(np.mean([out[k].n_correspondences for k in range(len(cands))])) if cands else 0.0
            rec = {
                "workload": what, "value": len(w.pairs) / (d / steps), "unit": "pairs/s",
                "ms_per_step": 1e3 * d / steps, "steps": steps, "pairs_per_step": len(w.pairs),
                "pairs_on_rank0": len(cands), "keyframes_on_rank0": len(keys),
                "points_per_keyframe": w.n_points, "submits_in_flight": 1,
                "pairs_ok_fraction_rank0": n_ok / max(len(cands), 1),
                "outer_iterations_mean_rank0": float(np.mean(its)) if its else 0.0,
                "mean_correspondences_rank0": tgt_pts,
                "registration_ms_per_keyframe": 1e3 * t_reg / max(len(keys), 1),
                "device_span_ms_last_submit_rank0": st.total_ms,
                "n_gpus": world if cut_over_ranks else 1,
                "rccl_ranks": (dist.get_world_size() if (cut_over_ranks and dist is not None) else None),
                "seed": w.seed,
            }
        finally:
            c.close()
        if exclusive_nn:
            # the lookup kernel's figures from the same batch on one stream (durations exclusive)
            cfg1 = capi.default_config(True, device=local_rank, init_method=capi.INIT_IDENTITY)
            cfg1.reserved[7] = 1
            cfg1.stage_timing = 1
            c1 = capi.Context(cfg1)
            try:
                register(c1, w, keys)
                pk1 = c1.pack_candidates(cands)
                c1.submit_packed(pk1)
                s0 = c1.stats()
                c1.submit_packed(pk1)
                s1 = c1.stats()
                launches = s1.nn_launches - s0.nn_launches
                flops = 8.0 * s1.nn_executed_evals / max(launches, 1)
                ms = s1.nn_ms / max(launches, 1)
                tf = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
                rec["nn_one_stream"] = {
                    "kernel": "lgicp::nn_pruned_kernel", "launches_per_step": launches,
                    "avg_launch_ms": ms, "flops_per_launch": flops, "achieved_tflops": tf,
                    "frac_of_fp32_valu_peak": tf / FP32_VALU_PEAK_TFLOPS,
                    "exhaustive_distances_per_launch": (s1.nn_distance_evals - s0.nn_distance_evals) / max(launches, 1),
                    "nn_ms_per_step": s1.nn_ms, "reduce_ms_per_step": s1.reduce_ms,
                    "solve_ms_per_step": s1.solve_ms, "prepare_ms_per_step": s1.cov_ms,
                    "device_span_ms": s1.total_ms}
            finally:
                c1.close()
        return rec

    subs = {}
    if args.config == "C2" and not args.no_sub_records and not strong and not args.exhaustive:
        ctx.close()
        ctx = None
        subs["accumulated"] = sub_record(
            "C2", "C2 with consecutive keys: the base-station block accumulates key +- 2 into every "
            "target (IcpLoopComputation.cc:910-947, laser_parameters.yaml:255-256), targets of up to "
            "5 x 16k points, built and indexed on the device the first time a candidate names them "
            "(lgicp_config.accumulated_cache_bytes) and found there by later submits", consecutive_keys=True,
            exclusive_nn=True)
        progress("sub-record: accumulated targets done")
        subs["solver_bfgs"] = sub_record(
            "C2", "C2 with lgicp_config.solver = LGICP_SOLVER_BFGS: the reference's optimiser on the "
            "reference's float objective, every function evaluation a pass over the correspondences",
            solver=capi.SOLVER_BFGS, steps=3, warm=1)
        progress("sub-record: BFGS back end done")
        subs["c3_list"] = sub_record(
            "C3", "C3, north_star's target configuration: ONE list of 1,000 candidate pairs over 250 "
            "keyframes of 32,768 points" + (f", cut over {world} GPUs (cost-balanced, keyframe affinity), "
                                            "result records gathered to rank 0 over RCCL inside the "
                                            "timed region" if world > 1 else ", whole on one GPU"),
            cut_over_ranks=True, steps=6, warm=2)
        progress("sub-record: C3 list done")
    if rank == 0:
        value = total_pairs_per_batch * args.steps / dt
        kernel_name = "lgicp::nn_kernel<4>" if args.exhaustive else "lgicp::nn_pruned_kernel"
        pmc, pmc_file = pmc_summary(args.config)
        src = excl["acc"] if excl else acc
        src_steps = excl["steps"] if excl else args.steps
        nn_launch_ms = src["nn_ms"] / max(src["nn_launches"], 1)
        algo_flops = 8.0 * src["nn_evals"] / max(src["nn_launches"], 1)
        exec_flops = (algo_flops if args.exhaustive
                      else 8.0 * src["nn_exec"] / max(src["nn_launches"], 1))
        achieved = exec_flops / (nn_launch_ms * 1e-3) / 1e12 if nn_launch_ms > 0 else 0.0
        red_bytes = 44.0 * src["red_points"] + 40.0 * src["red_inliers"]
        red_ms = src["red_ms"]
        red_gbs = red_bytes / (red_ms * 1e-3) / 1e9 if red_ms > 0 else 0.0
        k_nn, k_red = pmc_kernel(pmc, kernel_name), pmc_kernel(pmc, "lgicp::gicp_reduce_kernel")
        red_bytes_per_launch = red_bytes / max(src["red_launches"], 1)

        def issue_util(k):
            # VALU instructions issued per SIMD cycle: each wave instruction occupies its SIMD's
            # VALU for 4 cycles (64 lanes over 16 ALUs); 256 CUs x 4 SIMDs
            if "SQ_INSTS_VALU" not in k or "GRBM_GUI_ACTIVE" not in k:
                return None
            cyc = k["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0   # the counter sums the 8 XCDs
            return 4.0 * k["SQ_INSTS_VALU"]["mean_per_launch"] / (1024.0 * cyc) if cyc > 0 else None

        w0 = works[0]
        out = {
            "metric": "loop-closure GICP pairs verified/sec",
            "value": value,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "value_single_submit": (total_pairs_per_batch / serial) if serial else value,
            "parity": ("unpinned: the TEASER++ stage restates a library that is absent from the reference "
                       "tree and no reference test holds a value for it" if args.config == "C4" and
                       args.c4_init == "teaser" else
                       ("unpinned beyond the reference's TestSacInitialAlign case (FPFH values, SAC-IA draws)"
                        if args.config == "C4" else "oracle (see DESIGN.md section 3)")),
            "registration_ms_per_keyframe": 1e3 * reg_s / max(reg_kf, 1),
            "value_cold": cold["value"] if cold else None,
            "solver": "bfgs" if args.solver == "bfgs" else "gauss_newton",
            "rccl_ranks": dist.get_world_size() if dist is not None else None,
            "config": {
                "workload": (f"{args.config}: {len(w0.pairs)} candidate pairs "
                             f"{'in all, cut over the ranks' if strong else 'per GPU and step'}, "
                             f"{w0.n_points}-pt VLP-16-like keyframes ({w0.n_keyframes} distinct per "
                             f"batch, steps rotate through {n_batches} batch(es), {in_flight} step(s) in "
                             f"flight at once), GICP base-station "
                             f"parameters (corr_dist 0.5, <=200 outer iterations, tf_epsilon 1e-10), "
                             + (("SAC-IA" if args.c4_init == "sac" else "TEASER++")
                                + " initialisation from FPFH matches at Harris keypoints "
                                f"(radius {args.harris_radius}), frames of a pair up to 90 deg / 5 m "
                                "apart, odometry guess unaware of it, "
                                if args.config == "C4" else "identity initial guess, ")
                             + "POINT2PLANE covariance"),
                "pairs_per_step": total_pairs_per_batch,
                "points_per_keyframe": w0.n_points,
                "distinct_batches": n_batches,
                "keyframes_resident_per_gpu": reg_kf,
                "pairs_ok_fraction": ok / max(n_res, 1),
                "outer_iterations_mean": float(np.mean(iters)) if iters else 0.0,
                "outer_iterations_max": int(max(iters)) if iters else 0,
                "median_translation_error_m": float(np.median(terr)) if terr else None,
                "median_rotation_error_rad": float(np.median(rerr)) if rerr else None,
                "cycle_exits_per_step": acc["cycle_exits"] / args.steps,
                "seed": w0.seed,
                "parallelism": (f"one candidate list cut over {world} GPU(s) by estimated cost with "
                                f"keyframe affinity, RCCL gather of the result records to rank 0"
                                if strong else
                                (f"every one of {world} GPUs runs its own batches, RCCL gather to rank 0"
                                 if world > 1 else "single GPU")),
                "stage_ms_per_step": ({("prepare_and_init" if k == "cov" else k): v / args.steps
                                       for k, v in stage.items()} if main_stage_timing or not excl else
                                      {("prepare_and_init" if k == "cov" else k): v / excl["steps"]
                                       for k, v in excl["stage"].items()}),
                "stage_ms_source": ("timed region" if main_stage_timing or not excl else
                                    "exclusive one-stream phase (lgicp_config.stage_timing is off in the "
                                    "timed region: its events between the kernels cost a lone submit 5-7 %)"),
                "stage_ms_what": "hipEvent time per stage summed over the parts of a submit: nn = lookups, "
                                 "reduce / solve = linearise-reduce and 6x6 solve (or the BFGS back end's two "
                                 "kernels), prepare_and_init = scan accumulation, index of accumulated clouds, "
                                 "k-NN normals / covariances where configured and the whole feature-based "
                                 "initialisation (FEATURES, TEASERPP), post = fitness / Ap reductions",
                "submits_in_flight": in_flight,
                "one_submit_at_a_time": (None if serial is None else
                                         {"ms_per_step": 1e3 * serial,
                                          "value": total_pairs_per_batch / serial,
                                          "what": "the same steps submitted one after the other on the "
                                                  "context as lgicp_create leaves it (one lane, four parts "
                                                  "of a submit side by side), before lgicp_set_submit_lanes"}),
                "registration": {"keyframes": reg_kf, "seconds": reg_s,
                                 "what": "lgicp_add_keyframes + lgicp_wait_registered per batch: upload, "
                                         "k-d index, covariances, records, NormalizePCloud constants"},
                "cold_batch": cold,
            },
            "roofline": {
                "bound": "fp32_valu",
                "kernel": kernel_name,
                "achieved": achieved,
                "peak": FP32_VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP32_VALU_PEAK_TFLOPS,
                "traffic": k_nn.get("hbm_bytes_per_launch"),
                "avg_launch_ms": nn_launch_ms,
                "launches_per_step": src["nn_launches"] / src_steps,
                "flops_per_launch": exec_flops,
                "exhaustive_flops_per_launch": algo_flops,
                "pruning": exec_flops / algo_flops if algo_flops > 0 else None,
                "valu_issue_utilisation": issue_util(k_nn),
                "valu_issue_utilisation_by_pass_class": ({k: v["valu_issue_utilisation"] for k, v in
                                                          pmc["nn_pruned_kernel_by_pass_class"].items()}
                                                         if pmc and pmc.get("nn_pruned_kernel_by_pass_class")
                                                         else None),
                "exclusive": excl is not None,
                "exclusive_phase": ({"steps": excl["steps"], "ms_per_step": excl["ms_per_step"],
                                     "stage_ms_per_step": {("prepare_and_init" if k == "cov" else k):
                                                           v / excl["steps"]
                                                           for k, v in excl["stage"].items()}}
                                    if excl else None),
                "pmc_summary": pmc_file,
                "note": "achieved = 8 flop (3 sub, 3 mul, 2 add, unfused by contract) per distance the "
                        "kernel really evaluated (device counter) / launch duration by HIP events on "
                        "the launching stream, from a phase that runs the same batches on one stream "
                        "(durations exclusive; the timed region above uses up to four streams).  The "
                        "kernel answers every query exactly while evaluating the fraction `pruning` "
                        "of the N*M distances an exhaustive search computes; its instructions go to "
                        "box tests and certificates, so its FLOP rate is far from the 157.3 TFLOP/s "
                        "packed-FMA peak by design — valu_issue_utilisation (SQ_INSTS_VALU x 4 / "
                        "(1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), committed PMC pass) says how busy the vector "
                        "ALUs are.  The exhaustive kernel (--exhaustive) reaches 0.37 of the peak.",
            },
            "roofline_reduce": {
                "bound": "hbm",
                "kernel": "lgicp::gicp_reduce_kernel",
                "achieved": red_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": red_gbs / HBM_PEAK_GBS,
                "traffic": k_red.get("hbm_bytes_per_launch"),
                "traffic_over_algorithmic": (k_red["hbm_bytes_per_launch"] / red_bytes_per_launch
                                             if k_red.get("hbm_bytes_per_launch") and red_bytes_per_launch
                                             else None),
                "avg_launch_ms": red_ms / max(src["red_launches"], 1),
                "bytes_per_launch": red_bytes_per_launch,
                "exclusive": excl is not None,
                "note": "algorithmic bytes: 44 B per source point read (16 p + 4 index + 24 C1) + 40 B "
                        "per correspondence inside corr_dist (16 q + 24 C2), f32 minimum layout; "
                        "durations from the one-stream phase.",
            },
        }
        if args.config == "C4" and args.c4_init == "teaser":
            # the kernels the C4 step spends its time in belong to the TEASER++ initialisation
            k_cl = pmc_kernel(pmc, "lgicp::clique_starts_group_kernel")
            k_core = pmc_kernel(pmc, "lgicp::core_peel_group_kernel")

            # with two submits in flight the launch durations of one overlap the other's kernels:
            # the figures come from the phase that runs the same steps one at a time
            t_acc = ser_acc if ser_acc is not None else acc
            t_steps = ser_acc["steps"] if ser_acc is not None else args.steps

            def hbm_line(kernel, ms, nbytes, launches, k, note):
                # These two kernels are chains of dependent steps; the bytes their algorithm asks for
                # come mostly out of L2 / Infinity Cache.  `achieved` is therefore the HBM traffic the
                # committed PMC pass measured per launch over the launch duration, and the algorithmic
                # rate (served by the caches) is reported beside it, not as an HBM fraction.
                per = nbytes / max(launches, 1)
                avg_ms = ms / max(launches, 1)
                traffic = k.get("hbm_bytes_per_launch")
                hbm_gbs = traffic / (avg_ms * 1e-3) / 1e9 if (traffic and avg_ms > 0) else None
                return {"bound": "hbm", "kernel": kernel, "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": hbm_gbs / HBM_PEAK_GBS if hbm_gbs is not None else None, "traffic": traffic,
                        "traffic_over_algorithmic": (traffic / per if traffic and per else None),
                        "algorithmic_bytes_per_launch": per,
                        "algorithmic_gbs_cache_side": nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                        "avg_launch_ms": avg_ms, "launches_per_step": launches / t_steps,
                        "ms_per_step": ms / t_steps, "exclusive": False,
                        "pmc_summary": pmc_file, "note": note}
            out["roofline_gicp_nn"] = out.pop("roofline")
            out["roofline_gicp_reduce"] = out.pop("roofline_reduce")
            out["roofline"] = hbm_line(
                "lgicp::clique_starts_group_kernel", t_acc["clique_ms"], t_acc["clique_bytes"], t_acc["clique_launches"],
                k_cl, "TEASER++ greedy clique search, one wave per start, all pairs of a group per launch. "
                "algorithmic bytes: one row of the scale-consistency bit matrix (n/8 B) per start and per "
                "clique member a start tries (device counter); durations by HIP events on the launching "
                "stream; beside it, on a second stream, only the next group's matcher and graph kernels run "
                "(about a tenth of the step).  The rows of a graph are read again and again by its "
                "starts, so most of these bytes are served by L2 / Infinity Cache (see traffic); the kernel is "
                "bound by the chain ballot -> row fetch -> intersect -> count of every step, not by HBM.")
            out["roofline_core_numbers"] = hbm_line(
                "lgicp::core_peel_group_kernel", t_acc["core_ms"], t_acc["core_bytes"], t_acc["core_launches"],
                k_core, "core numbers by peeling, one workgroup per graph, one launch per group of pairs. "
                "algorithmic bytes: the bit matrix (n*n/8 B) of every graph once (a vertex's row is read when "
                "it leaves).  A graph is a chain of thousands of dependent steps (compare, barrier, row fetch, "
                "decrement), so a launch lasts as long as its longest chain at one memory latency per step: "
                "the kernel is bound by that latency, not by HBM bandwidth.")
        for k, v in subs.items():
            if v is not None:
                out[k] = v
        if not args.no_cpu and world == 1:
            cores = cpu_threads_available()
            if args.config == "C4":
                out["cpu_baseline"] = cpu_baseline_c4(w0, min(args.cpu_sample, cores, len(w0.pairs)),
                                                      cores, args.harris_radius, args.c4_init)
            else:
                out["cpu_baseline"] = cpu_baseline(w0, min(args.cpu_sample, len(w0.pairs)), cores)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if ctx is not None:
        ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
