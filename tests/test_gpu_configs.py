"""The benchmark configurations as parity cases: one pair of every keyframe size the configs name
(10k, 16k, 32k, 64k points) through lgicp_submit against the CPU oracle, and size-independent
properties of a larger batch (batch independence, slot independence, pose-inverse consistency)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _rot_err(Ta, Tb):
    # (angle, translation) of inv(Ta) @ Tb; the angle from sin and cos together (arccos of the
    # trace alone cannot resolve angles under ~3e-4 rad on float32 matrices)
    return O.pose_gap(Ta, Tb)


def _close_per_entry(got, ref, rtol):
    """Per-entry relative comparison of a 6x6 matrix; entries below 1e-9 of the largest one (sums
    that cancel to rounding noise) are compared against that floor instead."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return np.allclose(got, ref, rtol=rtol, atol=1e-9 * np.abs(ref).max())


def _oracle_pair(work, a, b, solver):
    src, sn = work.keyframes[a]
    tgt, tn = work.keyframes[b]
    params = O.base_icp_params(init_method=O.INIT_IDENTITY, solver=solver, threads=8,
                               transform_thresholding=0)
    ident = O.pose_from_quat([1, 0, 0, 0], [0, 0, 0])
    return O.perform_alignment(params, src, sn, tgt, tn, src, tn, ident)


@pytest.mark.parametrize("config,points", [("C1", None), ("C2", None), ("C3", None), ("C5", None)])
def test_one_pair_of_each_configuration(config, points):
    """north_star tolerances: pose 1e-4 rad / 1e-4 m, fitness and information 1e-5 relative per
    entry (against the oracle's Gauss-Newton back end; against its BFGS back end see
    tests/test_gpu_solver_gap.py, which measures 32 pairs per configuration)."""
    from lamp_amd import capi
    from lamp_amd.workload import Workload
    work = Workload(config, pairs=2, keyframes=2)
    c = capi.Context(init_method=capi.INIT_IDENTITY, transform_thresholding=0)
    try:
        work.register(c)
        res = [capi.result_to_dict(r) for r in c.submit(work.candidates())]
    finally:
        c.close()
    for d, (a, b) in zip(res, work.pairs):
        gn = _oracle_pair(work, a, b, O.SOLVER_GN)
        assert d["status"] == gn["status"] == 0
        ang, te = _rot_err(gn["T_icp"], d["T_icp"])
        assert ang < 1e-4 and te < 1e-4, (config, ang, te)
        assert abs(d["fitness"] - gn["fitness"]) <= 1e-5 * gn["fitness"]
        assert _close_per_entry(d["covariance"], gn["covariance"], 1e-4)
        src, sn = work.keyframes[a]
        tgt, tn = work.keyframes[b]
        _, idx = O.fitness_score(src, tgt, d["T_icp"])
        _, Ap = O.icp_cov_point_plane(src, tn, idx, d["T_icp"])
        assert _close_per_entry(d["information"], Ap / 0.0025, 1e-5)
        # the alignment is close to the synthetic truth (sanity, not parity)
        ang, te = _rot_err(work.truth[(a, b)], d["T_icp"])
        assert ang < 5e-3 and te < 2e-2
    if config == "C1":      # the reference's own solver on its CPU-runnable configuration
        a, b = work.pairs[0]
        bf = _oracle_pair(work, a, b, O.SOLVER_BFGS)
        ang, te = _rot_err(bf["T_icp"], res[0]["T_icp"])
        assert ang < 1e-3 and te < 1e-3


@pytest.mark.parametrize("solver", ["gn", "bfgs"])
@pytest.mark.parametrize("config", ["C1", "C2"])
def test_one_pair_of_each_configuration_with_knn_covariances(config, solver):
    """The covariance branch north_star's kernel (2) names: 20 nearest neighbours in the point's
    own cloud, SVD, singular values replaced by (1, 1, 1e-3) (gicp.hpp:83-155), instead of the
    stored normals.  Same tolerances as above, against the oracle run with the same branch; with
    the BFGS back end against the oracle's BFGS in the device's order of additions."""
    from lamp_amd import capi
    from lamp_amd.workload import Workload
    work = Workload(config, pairs=2, keyframes=2)
    c = capi.Context(init_method=capi.INIT_IDENTITY, transform_thresholding=0, covariance_from_knn=1,
                     solver=capi.SOLVER_BFGS if solver == "bfgs" else capi.SOLVER_GAUSS_NEWTON)
    try:
        work.register(c)
        res = [capi.result_to_dict(r) for r in c.submit(work.candidates())]
    finally:
        c.close()
    ident = O.pose_from_quat([1, 0, 0, 0], [0, 0, 0])
    for d, (a, b) in zip(res, work.pairs):
        src, sn = work.keyframes[a]
        tgt, tn = work.keyframes[b]
        params = O.base_icp_params(init_method=O.INIT_IDENTITY, threads=8, transform_thresholding=0,
                                   cov_from_knn=1,
                                   solver=O.SOLVER_BFGS if solver == "bfgs" else O.SOLVER_GN,
                                   sum_order=1 if solver == "bfgs" else 0)
        ref = O.perform_alignment(params, src, sn, tgt, tn, src, tn, ident)
        assert d["status"] == ref["status"] == 0
        ang, te = _rot_err(ref["T_icp"], d["T_icp"])
        assert ang < 1e-4 and te < 1e-4, (config, solver, ang, te)
        assert abs(d["fitness"] - ref["fitness"]) <= 1e-5 * ref["fitness"]
        assert _close_per_entry(d["covariance"], ref["covariance"], 1e-4)
        _, idx = O.fitness_score(src, tgt, d["T_icp"])
        _, Ap = O.icp_cov_point_plane(src, tn, idx, d["T_icp"])
        assert _close_per_entry(d["information"], Ap / 0.0025, 1e-5)
        ang, te = _rot_err(work.truth[(a, b)], d["T_icp"])
        assert ang < 5e-3 and te < 2e-2


def test_accumulated_cloud_cache_changes_nothing():
    """Base-station parameters accumulate key +- 2 into every target (IcpLoopComputation.cc:910-947).
    The engine keeps such clouds on the device with a search index of their own once a submit has
    built them (lgicp_config.accumulated_cache_bytes): the results are those of a context that
    assembles them afresh in every submit, bit for bit, also when a neighbour scan arrives between
    two submits and the cloud has to be rebuilt with it."""
    from lamp_amd import capi
    from lamp_amd.workload import Workload
    work = Workload("C2", pairs=48, points=4096, keyframes=16, consecutive_keys=True)
    keys = sorted(work.keyframes)
    late = keys[3]                                   # a neighbour of several targets

    def run(cache_bytes, lanes=1):
        c = capi.Context(init_method=capi.INIT_IDENTITY, accumulated_cache_bytes=cache_bytes)
        try:
            first = [k for k in keys if k != late]
            c.add_keyframes(first, [work.keyframes[k][0] for k in first], [work.keyframes[k][1] for k in first])
            for k in keys:
                c.add_pose(k, *work.poses[k])
            cands = [p for p in work.candidates() if late not in p]
            outs = [[capi.result_to_dict(r) for r in c.submit(cands)] for _ in range(2)]
            c.add_keyframe(late, *work.keyframes[late])   # targets next to it now hold one scan more
            outs += [[capi.result_to_dict(r) for r in c.submit(cands)] for _ in range(2)]
            return outs
        finally:
            c.close()

    cached, fresh = run(0), run(1)
    # a budget of three or four clouds for sixteen targets: entries are evicted and rebuilt all the time
    tight = run(16 << 20)
    for a_list, b_list in zip(cached, tight):
        for a, b in zip(a_list, b_list):
            assert a["status"] == b["status"] and np.array_equal(a["T_icp"], b["T_icp"])
            assert a["fitness"] == b["fitness"]
    for a_list, b_list in zip(cached, fresh):
        for a, b in zip(a_list, b_list):
            assert a["status"] == b["status"] and a["iterations"] == b["iterations"]
            assert np.array_equal(a["T_icp"], b["T_icp"]) and a["fitness"] == b["fitness"]
            assert np.array_equal(a["information"], b["information"])
            assert a["n_correspondences"] == b["n_correspondences"]
    # a submit that finds its clouds cached returns what the submit that built them returned
    for a, b in zip(cached[0], cached[1]):
        assert np.array_equal(a["T_icp"], b["T_icp"]) and a["fitness"] == b["fitness"]
    # the late scan did change some targets (the test would be vacuous otherwise)
    assert any(not np.array_equal(a["T_icp"], b["T_icp"]) for a, b in zip(cached[1], cached[2]))
    assert sum(r["status"] == 0 for r in cached[3]) >= 0.8 * len(cached[3])


@pytest.mark.parametrize("solver", ["gn", "bfgs"])
def test_accumulated_targets_with_submits_side_by_side(solver):
    """Several host threads submit the same accumulated targets at once on a fresh context (so
    that more than one of them finds a cloud missing and builds it): every submit returns what a
    lone submit on a context without the cache returns, with either inner solver."""
    import threading
    from lamp_amd import capi
    from lamp_amd.workload import Workload
    work = Workload("C2", pairs=64, points=4096, keyframes=16, consecutive_keys=True)
    sv = capi.SOLVER_BFGS if solver == "bfgs" else capi.SOLVER_GAUSS_NEWTON

    def to_dicts(res):
        return [capi.result_to_dict(r) for r in res]

    c0 = capi.Context(init_method=capi.INIT_IDENTITY, accumulated_cache_bytes=1, solver=sv)
    try:
        work.register(c0)
        ref = to_dicts(c0.submit(work.candidates()))
    finally:
        c0.close()
    c = capi.Context(init_method=capi.INIT_IDENTITY, solver=sv)
    try:
        work.register(c)
        assert c.set_submit_lanes(4) == 4
        halves = [work.candidates()[:40], work.candidates()[24:]]       # overlapping target sets
        got, errors = {}, []

        def submitter(k):
            try:
                for rep in range(3):
                    got[(k, rep)] = to_dicts(c.submit(halves[k % 2]))
            except Exception as e:       # noqa: BLE001
                errors.append(e)

        th = [threading.Thread(target=submitter, args=(k,)) for k in range(4)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors
        for (k, rep), res in got.items():
            want = ref[:40] if k % 2 == 0 else ref[24:]
            for a, b in zip(res, want):
                assert a["status"] == b["status"] and a["iterations"] == b["iterations"]
                assert np.array_equal(a["T_icp"], b["T_icp"]) and a["fitness"] == b["fitness"]
                assert np.array_equal(a["information"], b["information"])
    finally:
        c.close()


@pytest.mark.parametrize("method", ["features", "teaser"])
def test_feature_initialisation_on_accumulated_targets(method):
    """Keypoints and descriptors of an accumulated target are those of the whole accumulated
    cloud (IcpLoopComputation.cc:436-447 runs after AccumulateScans).  With the cloud cached on the
    device they are computed once and kept beside it: initial alignment and refined pose equal
    those of a context that assembles the cloud and its features afresh in every submit."""
    from lamp_amd import capi
    from lamp_amd.workload import Workload
    work = Workload("C4", pairs=8, points=4096, keyframes=8, consecutive_keys=True)
    init = capi.INIT_FEATURES if method == "features" else capi.INIT_TEASERPP

    def run(cache_bytes):
        c = capi.Context(init_method=init, harris_radius=0.5, transform_thresholding=0,
                         accumulated_cache_bytes=cache_bytes)
        try:
            work.register(c)
            return [[capi.result_to_dict(r) for r in c.submit(work.candidates())] for _ in range(2)]
        finally:
            c.close()

    cached, fresh = run(0), run(1)
    for a_list, b_list in zip(cached, fresh):
        for a, b in zip(a_list, b_list):
            assert a["status"] == b["status"] and a["iterations"] == b["iterations"]
            assert np.array_equal(a["T_init"], b["T_init"]) and np.array_equal(a["T_icp"], b["T_icp"])
            assert a["fitness"] == b["fitness"] and a["init_fitness"] == b["init_fitness"]
    for a, b in zip(cached[0], cached[1]):
        assert np.array_equal(a["T_init"], b["T_init"]) and np.array_equal(a["T_icp"], b["T_icp"])
    # the initial alignment did something: it differs from the odometry guess for most pairs
    assert sum(r["init_fitness"] > 0 for r in cached[0]) >= 4


def test_stage_timing_is_optional_and_changes_nothing():
    """lgicp_config.stage_timing puts events between the kernels of every outer iteration to fill the
    per-stage times of lgicp_stats; without it only the device span of the submit is measured.
    The results are the same bits."""
    from lamp_amd import capi
    from lamp_amd.workload import Workload
    work = Workload("C2", pairs=32, points=4096, keyframes=16)
    outs, stats = [], []
    for timing in (0, 1):
        c = capi.Context(init_method=capi.INIT_IDENTITY, stage_timing=timing)
        try:
            work.register(c)
            res, st = c.submit_packed_stats(c.pack_candidates(work.candidates()))
            outs.append([capi.result_to_dict(res[k]) for k in range(len(work.pairs))])
            stats.append(st)
        finally:
            c.close()
    for a, b in zip(*outs):
        assert a["status"] == b["status"] and a["iterations"] == b["iterations"]
        assert np.array_equal(a["T_icp"], b["T_icp"]) and a["fitness"] == b["fitness"]
        assert np.array_equal(a["information"], b["information"])
    assert stats[0].nn_ms == 0 and stats[0].reduce_ms == 0 and stats[0].solve_ms == 0
    assert stats[0].total_ms > 0 and stats[0].nn_launches == stats[1].nn_launches > 0
    assert stats[1].nn_ms > 0 and stats[1].reduce_ms > 0 and stats[1].solve_ms > 0
    assert stats[1].nn_ms + stats[1].reduce_ms + stats[1].solve_ms <= 4.0 * stats[1].total_ms   # four parts side by side


def test_batch_and_slot_independence():
    """A pair's result does not depend on what else is in the submit, on the order of the
    candidates, or on how the submit is cut over the two concurrent batch slots."""
    from lamp_amd import capi
    from lamp_amd.workload import Workload
    work = Workload("C2", pairs=96, points=4096, keyframes=16)
    cands = work.candidates()

    def run(cand_list, slots):
        res = [0] * 8
        res[7] = slots
        c = capi.Context(init_method=capi.INIT_IDENTITY, reserved=(capi.C.c_int * 8)(*res))
        try:
            work.register(c)
            return [capi.result_to_dict(r) for r in c.submit(cand_list)]
        finally:
            c.close()

    two = run(cands, 2)
    one = run(cands, 1)
    rev = run(cands[::-1], 2)[::-1]
    solo = run(cands[5:6], 1)
    for a, b, r in zip(two, one, rev):
        for other in (b, r):
            assert a["status"] == other["status"] and a["iterations"] == other["iterations"]
            assert np.array_equal(a["T_icp"], other["T_icp"])
            assert a["fitness"] == other["fitness"]
            assert np.array_equal(a["information"], other["information"])
    assert np.array_equal(solo[0]["T_icp"], two[5]["T_icp"])
    assert sum(r["status"] == 0 for r in two) >= 90
    # edge transform = inverse of the ICP transform (gu::PoseInverse, IcpLoopComputation.cc:617)
    for r in two:
        if r["status"] == 0:
            T = np.asarray(r["T_icp"], np.float64)
            assert np.allclose(r["delta_R"] @ T[:3, :3], np.eye(3), atol=1e-6)
            assert np.allclose(r["delta_R"] @ T[:3, 3] + r["delta_t"], 0, atol=1e-6)


@pytest.mark.parametrize("lanes,threads", [(2, 2), (4, 4), (2, 5), (6, 6)])
def test_submits_side_by_side_equal_submits_in_turn(lanes, threads):
    """lgicp_set_submit_lanes: submits issued by several host threads at once (more threads than
    lanes wait their turn) return what the same submits return one after the other, bit for bit;
    a registration issued meanwhile waits for the submits in flight, and the keyframes it adds are
    there for the next submit."""
    import threading
    from lamp_amd import capi
    from lamp_amd.workload import Workload
    works = [Workload("C2", pairs=48, points=4096, keyframes=8, shard=s, key_offset=100000 * s)
             for s in range(threads)]
    late = Workload("C2", pairs=8, points=4096, keyframes=4, shard=17, key_offset=9000000)

    def to_dicts(res):
        return [capi.result_to_dict(r) for r in res]

    c = capi.Context(init_method=capi.INIT_IDENTITY)
    try:
        for w in works:
            w.register(c)
        serial = [to_dicts(c.submit(w.candidates())) for w in works]
        assert c.set_submit_lanes(lanes) == lanes
        got, errors = [None] * threads, []

        def submitter(k):
            try:
                for _ in range(3):
                    got[k] = to_dicts(c.submit(works[k].candidates()))
            except Exception as e:   # noqa: BLE001 - reported by the main thread
                errors.append(e)

        def registrar():
            try:
                late.register(c)
            except Exception as e:   # noqa: BLE001
                errors.append(e)

        th = [threading.Thread(target=submitter, args=(k,)) for k in range(threads)]
        th.append(threading.Thread(target=registrar))
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors
        for a_list, b_list in zip(serial, got):
            for a, b in zip(a_list, b_list):
                assert a["status"] == b["status"] and a["iterations"] == b["iterations"]
                assert np.array_equal(a["T_icp"], b["T_icp"]) and a["fitness"] == b["fitness"]
                assert np.array_equal(a["information"], b["information"])
        res = to_dicts(c.submit(late.candidates()))
        assert sum(r["status"] == 0 for r in res) >= 6
        assert c.set_submit_lanes(1) == 1
        again = to_dicts(c.submit(works[0].candidates()))
        assert all(np.array_equal(a["T_icp"], b["T_icp"]) for a, b in zip(serial[0], again))
    finally:
        c.close()


def test_registration_is_not_starved_by_overlapping_submits():
    """Six threads submit without a pause; a registration that arrives meanwhile goes ahead of the
    submits that arrive after it instead of waiting for a gap that never comes."""
    import threading
    import time
    from lamp_amd import capi
    from lamp_amd.workload import Workload
    works = [Workload("C2", pairs=48, points=4096, keyframes=8, shard=s, key_offset=100000 * s)
             for s in range(6)]
    late = Workload("C2", pairs=8, points=4096, keyframes=4, shard=17, key_offset=9000000)
    c = capi.Context(init_method=capi.INIT_IDENTITY)
    try:
        for w in works:
            w.register(c)
        assert c.set_submit_lanes(6) == 6
        stop, errors, counts = threading.Event(), [], [0] * 6

        def submitter(k):
            try:
                pk = c.pack_candidates(works[k].candidates())
                while not stop.is_set():
                    c.submit_packed(pk)
                    counts[k] += 1
            except Exception as e:   # noqa: BLE001 - reported by the main thread
                errors.append(e)

        th = [threading.Thread(target=submitter, args=(k,)) for k in range(6)]
        for t in th:
            t.start()
        time.sleep(0.3)
        before = sum(counts)
        t0 = time.perf_counter()
        late.register(c)
        waited = time.perf_counter() - t0
        time.sleep(0.2)
        stop.set()
        for t in th:
            t.join()
        assert not errors, errors
        assert before > 0 and sum(counts) > before     # submits ran before and after it
        assert waited < 1.0, waited
        res = c.submit(late.candidates())
        assert sum(r.status == 0 for r in res) >= 6
    finally:
        c.close()


def test_small_workspace_cuts_the_submit_into_sub_batches():
    """workspace_bytes bounds the per-batch scratch (configuration C5 does not fit one batch):
    the results do not depend on where the cuts fall."""
    from lamp_amd import capi
    from lamp_amd.workload import Workload
    work = Workload("C2", pairs=40, points=4096, keyframes=8)
    outs = []
    for ws in (0, 600_000):                       # default 16 GiB, then room for ~3 pairs
        c = capi.Context(init_method=capi.INIT_IDENTITY, workspace_bytes=ws)
        try:
            work.register(c)
            outs.append([capi.result_to_dict(r) for r in c.submit(work.candidates())])
        finally:
            c.close()
    for a, b in zip(*outs):
        assert a["status"] == b["status"] and a["iterations"] == b["iterations"]
        assert np.array_equal(a["T_icp"], b["T_icp"]) and a["fitness"] == b["fitness"]


def test_robot_parameter_block():
    """The reference's robot-side block (laser_parameters.yaml:5-147): corr_dist 0.1, 5 outer
    iterations, fitness gate 0.4, no transform thresholding — pairs mostly stop at the
    iteration limit, which counts as converged (gicp.hpp:564-580)."""
    from lamp_amd import capi
    from lamp_amd.workload import Workload
    work = Workload("C1", pairs=2, keyframes=2)
    cfg = capi.default_config(False, init_method=capi.INIT_IDENTITY, accumulate_source=0)
    c = capi.Context(cfg)
    try:
        work.register(c)
        res = [capi.result_to_dict(r) for r in c.submit(work.candidates())]
    finally:
        c.close()
    for d, (a, b) in zip(res, work.pairs):
        src, sn = work.keyframes[a]
        tgt, tn = work.keyframes[b]
        params = O.base_icp_params(init_method=O.INIT_IDENTITY, solver=O.SOLVER_GN, threads=8,
                                   corr_dist=0.1, max_iterations=5, max_tolerable_fitness=0.4,
                                   transform_thresholding=0)
        ref = O.perform_alignment(params, src, sn, tgt, tn, src, tn,
                                  O.pose_from_quat([1, 0, 0, 0], [0, 0, 0]))
        assert d["status"] == ref["status"]
        assert d["iterations"] == ref["iterations"] <= 5
        ang, te = _rot_err(ref["T_icp"], d["T_icp"])
        assert ang < 1e-4 and te < 1e-4
        assert abs(d["fitness"] - ref["fitness"]) <= 1e-5 * ref["fitness"]
