"""Configuration C4: FPFH + TEASER++ (and SAC-IA) global registration of pairs whose frames are up
to 90 degrees / 5 m apart while their registered odometry poses know nothing of it
(IcpLoopComputation::GetTeaserInitialAlignment, loop_closure/src/IcpLoopComputation.cc:949-1081;
GetSacInitialAlignment :650-705).  Four 16k-point pairs per method against the CPU oracle, then a
larger batch checked through a property: the recovered pose is the synthetic truth.

TEASER++ is not in the reference tree and its test there is commented out
(test_loop_computation.cc:101-123): the oracle restates the publication, parity unpinned."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

RADIUS = 0.5   # Harris radius the C4 bench line uses on these ring-sampled 16k-point scans


def _odom_guess(work, a, b):
    p21 = O.pose_between(O.pose_from_quat(*work.poses[b]), O.pose_from_quat(*work.poses[a]))
    g = np.eye(4, dtype=np.float32)
    g[:3, :3] = np.array(p21.R).reshape(3, 3)
    g[:3, 3] = np.array(p21.t)
    return g


def _submit(work, method, radius=RADIUS, **kw):
    from lamp_amd import capi
    c = capi.Context(init_method=method, harris_radius=radius, transform_thresholding=0, **kw)
    try:
        work.register(c)
        return [capi.result_to_dict(r) for r in c.submit(work.candidates())]
    finally:
        c.close()


def test_c4_workload_has_large_misalignment():
    from lamp_amd.workload import Workload
    work = Workload("C4", pairs=64, keyframes=16)
    rot, tr, guess_err = [], [], []
    for a, b in work.pairs:
        T = work.truth[(a, b)]
        ang, te = O.pose_gap(np.eye(4), T)
        rot.append(ang), tr.append(te)
        guess_err.append(O.pose_gap(_odom_guess(work, a, b), T))
    assert np.rad2deg(max(rot)) <= 90.0 + 1e-6 and max(tr) <= 5.0 + 0.6
    assert np.rad2deg(np.median(rot)) > 15.0 and np.median(tr) > 0.8
    # the odometry guess is off by about as much: nothing GICP's 0.5 m gate could bridge
    assert np.median([g[0] for g in guess_err]) > np.deg2rad(15.0)
    assert np.median([g[1] for g in guess_err]) > 0.8


def test_c4_teaser_pairs_match_oracle():
    from lamp_amd import capi
    from lamp_amd.workload import Workload
    work = Workload("C4", pairs=4, keyframes=4)
    assert work.n_points == 16384
    res = _submit(work, capi.INIT_TEASERPP)
    for d, (a, b) in zip(res, work.pairs):
        src, sn = work.keyframes[a]
        tgt, tn = work.keyframes[b]
        g = _odom_guess(work, a, b)
        T0, used, nm, nc = O.teaser_initial_alignment(src, sn, tgt, tn, harris_radius=RADIUS, guess=g)
        assert nm > 10
        assert np.allclose(d["T_init"], T0, atol=1e-6), (a, b)
        assert d["init_fitness"] == (nc if used else 0)
        ref = O.gicp_align(O.base_gicp_params(solver=O.SOLVER_GN), src, sn, tgt, tn, guess=T0)
        assert d["converged"] == ref["converged"]
        ang, te = O.pose_gap(ref["T"], d["T_icp"])
        assert ang < 1e-4 and te < 1e-4, (a, b, ang, te)
        assert d["iterations"] == ref["iterations"]


def test_c4_sac_pairs_match_oracle():
    from lamp_amd import capi
    from lamp_amd.workload import Workload
    work = Workload("C4", pairs=4, keyframes=4)
    res = _submit(work, capi.INIT_FEATURES)
    for d, (a, b) in zip(res, work.pairs):
        src, sn = work.keyframes[a]
        tgt, tn = work.keyframes[b]
        g = _odom_guess(work, a, b)
        T0, fit0, _ = O.sac_initial_alignment(src, sn, tgt, tn, harris_radius=RADIUS, iterations=500,
                                              guess=g, seed=capi.pair_seed(a, b))
        assert np.allclose(d["T_init"], T0, atol=1e-5), (a, b)
        assert abs(d["init_fitness"] - fit0) <= 1e-5 * max(fit0, 1e-12)
        if not fit0 < 32.0:      # sac_fitness_score_threshold of the base block
            assert d["status"] == capi.PAIR_INIT_REJECTED
            continue
        ref = O.gicp_align(O.base_gicp_params(solver=O.SOLVER_GN), src, sn, tgt, tn, guess=T0)
        assert d["converged"] == ref["converged"]
        ang, te = O.pose_gap(ref["T"], d["T_icp"])
        assert ang < 1e-4 and te < 1e-4, (a, b, ang, te)


@pytest.mark.parametrize("method,min_recovered", [("teaser", 0.8), ("sac", 0.75)])
def test_c4_batch_recovers_the_true_poses(method, min_recovered):
    """Size-independent property on a larger batch (64 pairs over 16 keyframes): the pose that
    comes out is the synthetic truth, although the guess is wrong by tens of degrees and metres."""
    from lamp_amd import capi
    from lamp_amd.workload import Workload
    work = Workload("C4", pairs=64, keyframes=16)
    # Harris radius 0.3: measured 58 (TEASER++) and 56 (SAC-IA) of these 64 pairs recovered; 0.5, the
    # radius of the oracle comparisons above, recovers 50 and 52 at a fifth of the matches
    res = _submit(work, capi.INIT_TEASERPP if method == "teaser" else capi.INIT_FEATURES, radius=0.3)
    good = 0
    for d, (a, b) in zip(res, work.pairs):
        ang, te = O.pose_gap(work.truth[(a, b)], d["T_icp"])
        good += d["status"] == 0 and ang < 0.02 and te < 0.1
    assert good >= min_recovered * len(work.pairs), (method, good)
    # and GICP alone, from the odometry guess, recovers next to none of them
    if method == "sac":
        return
    lost = _submit(work, capi.INIT_ODOMETRY)
    n_lost = sum(not (d["status"] == 0 and O.pose_gap(work.truth[p], d["T_icp"])[1] < 0.1)
                 for d, p in zip(lost, work.pairs))
    assert n_lost >= 0.8 * len(work.pairs), n_lost


def test_c4_group_results_do_not_depend_on_the_grouping():
    """The TEASER++ stage advances all pairs of a submit together (one launch carries a step of
    every pair's clique search, core numbers and rotation search).  What a pair gets must not depend
    on its company: twelve pairs in one submit against the same pairs one per submit, bit for bit."""
    from lamp_amd import capi
    from lamp_amd.workload import Workload
    work = Workload("C4", pairs=12, keyframes=6, points=8192)
    c = capi.Context(init_method=capi.INIT_TEASERPP, harris_radius=0.4, transform_thresholding=0)
    try:
        work.register(c)
        together = [capi.result_to_dict(r) for r in c.submit(work.candidates())]
        alone = [capi.result_to_dict(c.submit([cand])[0]) for cand in work.candidates()]
    finally:
        c.close()
    assert len({d["init_fitness"] for d in together}) > 1      # cliques of different sizes in one group
    for a, b in zip(together, alone):
        assert a["init_fitness"] == b["init_fitness"]
        assert np.array_equal(a["T_init"], b["T_init"])
        assert np.array_equal(a["T_icp"], b["T_icp"]) and a["status"] == b["status"]


def test_c4_large_submit_goes_through_several_groups():
    """A TEASERPP submit takes its pairs in groups (the 64 with the most matches first, then up to
    512 at a time, each group prepared on a helper thread beside the search of the one before).  600
    pairs are three groups; cut into submits of 50, which are one group each, they must come out the
    same, bit for bit."""
    from lamp_amd import capi
    from lamp_amd.workload import Workload
    work = Workload("C4", pairs=600, keyframes=16, points=2048)
    cands = work.candidates()
    c = capi.Context(init_method=capi.INIT_TEASERPP, harris_radius=0.6, transform_thresholding=0)
    try:
        work.register(c)
        whole = [capi.result_to_dict(r) for r in c.submit(cands)]
        parts = []
        for k in range(0, len(cands), 50):
            parts += [capi.result_to_dict(r) for r in c.submit(cands[k:k + 50])]
    finally:
        c.close()
    assert len(whole) == len(parts) == 600
    assert sum(d["init_fitness"] > 0 for d in whole) >= 300     # TEASER++ was used, not just the guess
    for a, b in zip(whole, parts):
        assert a["init_fitness"] == b["init_fitness"] and a["status"] == b["status"]
        assert np.array_equal(a["T_init"], b["T_init"]) and np.array_equal(a["T_icp"], b["T_icp"])


@pytest.mark.parametrize("method", ["TEASERPP", "FEATURES"])
def test_c4_submits_side_by_side_equal_submits_in_turn(method):
    """A context with a feature-based initialisation takes two submits at a time
    (lgicp_set_submit_lanes: each works with its own set of feature buffers and streams).  Three
    threads, each with its own batch over keyframes the batches share, the first submits computing
    the keyframes' features: every result equals the one the batch got when submitted alone."""
    import threading
    from lamp_amd import capi
    from lamp_amd.workload import Workload
    work = Workload("C4", pairs=36, keyframes=8, points=8192)
    cands = work.candidates()
    batches = [cands[0:12], cands[12:24], cands[24:36]]

    def dicts(res):
        return [capi.result_to_dict(r) for r in res]

    def make():
        c = capi.Context(init_method=getattr(capi, "INIT_" + method), harris_radius=0.4,
                         transform_thresholding=0)
        work.register(c)
        return c

    ref_ctx = make()
    try:
        ref = [dicts(ref_ctx.submit(b)) for b in batches]
    finally:
        ref_ctx.close()
    c = make()                       # fresh: no keyframe has features yet
    try:
        assert c.set_submit_lanes(4) == 2
        got, errors = [None] * 3, []

        def run(k):
            try:
                for _ in range(2):
                    got[k] = dicts(c.submit(batches[k]))
            except Exception as e:   # noqa: BLE001 - reported by the main thread
                errors.append(e)

        th = [threading.Thread(target=run, args=(k,)) for k in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors
    finally:
        c.close()
    for a_list, b_list in zip(ref, got):
        for a, b in zip(a_list, b_list):
            assert a["status"] == b["status"] and a["init_fitness"] == b["init_fitness"]
            assert np.array_equal(a["T_init"], b["T_init"]) and np.array_equal(a["T_icp"], b["T_icp"])


@pytest.mark.parametrize("method", ["features", "teaser"])
def test_feature_sweeps_skip_only_what_is_out_of_reach(method):
    """The radius sweeps of Harris3D and FPFH skip the 64-point stretches of a scan whose box lies
    farther away than the radius.  With that switched off (reserved[5] bit 2: every point is looked
    at) a submit returns the same bits: keypoints, descriptors, initial alignment, refined pose."""
    from lamp_amd import capi
    from lamp_amd.workload import Workload
    work = Workload("C4", pairs=8, points=8192, keyframes=8)
    init = capi.INIT_FEATURES if method == "features" else capi.INIT_TEASERPP
    outs = []
    for bits in (0, 4):
        res = [0] * 8
        res[5] = bits
        c = capi.Context(init_method=init, harris_radius=RADIUS, transform_thresholding=0,
                         reserved=(capi.C.c_int * 8)(*res))
        try:
            work.register(c)
            outs.append([capi.result_to_dict(r) for r in c.submit(work.candidates())])
        finally:
            c.close()
    assert sum(r["init_fitness"] > 0 for r in outs[0]) >= 4      # the initialisation did run
    for a, b in zip(*outs):
        assert a["status"] == b["status"] and a["iterations"] == b["iterations"]
        assert a["init_fitness"] == b["init_fitness"]
        assert np.array_equal(a["T_init"], b["T_init"]) and np.array_equal(a["T_icp"], b["T_icp"])
        assert a["fitness"] == b["fitness"]

