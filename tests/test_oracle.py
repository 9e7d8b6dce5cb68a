"""CPU oracle pinned against the known answers held by the reference's own tests
(lamp_utils/test/test_point_cloud_utils.cc, loop_closure/test/test_loop_computation.cc)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import fixtures as F

TOL = 1e-5  # tolerance_ of the reference fixtures


def test_plane_normals_fixture():
    # test_point_cloud_utils.cc:55-67 (ExtractNormals on the plane)
    xyz, nrm = F.generate_plane()
    assert xyz.shape == (100, 3)
    assert np.allclose(nrm, [0, 0, 1], atol=TOL)


def test_normalize_pcloud():
    # test_point_cloud_utils.cc:87-112
    xyz, _ = F.generate_plane()
    q = O.normalize_cloud(xyz).astype(np.float64)
    assert len(q) == 100
    assert np.allclose(q.mean(0), 0, atol=TOL)
    assert abs(np.sqrt((q * q).sum(1)).mean() - 1) < TOL


def test_compute_ap_known_values():
    # test_point_cloud_utils.cc:183-231
    xyz, nrm = F.generate_plane()
    qn = O.normalize_cloud(xyz)
    corr = np.arange(100)
    Ap = O.compute_ap(qn, nrm, corr, np.eye(4))
    ref = np.zeros((6, 6))
    for i in range(100):
        a = np.cross(qn[i].astype(np.float64), nrm[i].astype(np.float64))
        c = float(nrm[i, 2])
        ref[0, 0] += a[0] * a[0]
        ref[0, 1] += a[0] * a[1]
        ref[1, 0] += a[0] * a[1]
        ref[1, 1] += a[1] * a[1]
        ref[0, 5] += a[0] * c
        ref[1, 5] += a[1] * c
        ref[5, 0] += a[0] * c
        ref[5, 1] += a[1] * c
        ref[5, 5] += c * c
    assert np.allclose(Ap, ref, atol=TOL)
    assert abs(Ap[0, 0] - 56.77534) < TOL
    assert abs(Ap[1, 1] - 56.77534) < TOL
    assert abs(Ap[5, 5] - 100) < TOL


def test_icp_observability():
    # test_point_cloud_utils.cc:172-181
    xyz, nrm = F.generate_plane()
    e = O.icp_observability(xyz, nrm)
    assert abs(e[0]) < TOL and abs(e[1]) < TOL and abs(e[2] - 100) < TOL


def test_compute_ap_out_of_range_normals_are_zero():
    # PointCloudUtils.cc:138-146: correspondences beyond the normal cloud contribute nothing
    xyz, nrm = F.generate_plane()
    qn = O.normalize_cloud(xyz)
    corr = np.arange(100)
    corr[50:] += 1000
    Ap = O.compute_ap(qn, nrm, corr, np.eye(4))
    assert abs(Ap[5, 5] - 50) < TOL


@pytest.mark.parametrize("solver", [O.SOLVER_BFGS, O.SOLVER_GN])
def test_perform_alignment_corner(solver):
    # test_loop_computation.cc:125-182: keys a100 (moved corner, source) -> a0 (corner, target),
    # perturbed odometry poses, expects delta ~= T = (1, -0.001, 0) to 1e-3
    corner, cn = F.generate_corner()
    moved, mn, T = F.shifted_corner()
    p0 = O.pose_from_quat([0, 0, 0, 0], [0, 0, 0.1])        # default-initialised quaternion
    p100 = O.pose_from_quat([0, 0, 0, 0], [-0.9, 0.1, 0])
    pose_21 = O.pose_between(p0, p100)                        # keyed_poses_[key2].between([key1])
    R21, t21 = O.pose_to_np(pose_21)
    assert np.allclose(R21, np.eye(3)) and np.allclose(t21, [-0.9, 0.1, -0.1])
    params = O.base_icp_params(init_method=O.INIT_ODOMETRY, solver=solver)
    r = O.perform_alignment(params, moved, mn, corner, cn, moved, cn, pose_21)
    assert r["status"] == 0 and r["converged"]
    assert np.allclose(r["delta_R"], np.eye(3), atol=1e-3)
    assert np.allclose(r["delta_t"], [1, -0.001, 0], atol=1e-3)
    assert r["fitness"] < 1e-6
    # covariance must be SPD for gtsam's Gaussian::Covariance (CommonFunctions.h:62-83)
    assert np.all(np.linalg.eigvalsh(0.5 * (r["covariance"] + r["covariance"].T)) > 0)


def test_threaded_equals_serial_signs():
    # test_icp_computation.cpp:53-162: corner at x=+-1, 100 iterations; edge sign must follow key parity
    corner, cn = F.generate_corner()
    params = O.base_icp_params(init_method=O.INIT_ODOMETRY, max_iterations=100)
    signs = []
    for k in range(4):
        k_to, k_from = k, k + 100
        shift = lambda kk: np.float32(-1.0 if kk % 2 == 1 else 1.0)
        tgt = corner.copy(); tgt[:, 0] = tgt[:, 0] + shift(k_to)
        src = corner.copy(); src[:, 0] = src[:, 0] + shift(k_from)
        pose = lambda kk: O.pose_from_quat([0, 0, 0, 0], [1.0 if kk % 2 == 1 else -1.0, 0, 0])
        pose_21 = O.pose_between(pose(k_to), pose(k_from))
        for threads in (1, 4):
            params.gicp.threads = threads
            r = O.perform_alignment(params, src, cn, tgt, cn, src, cn, pose_21)
            assert r["status"] == 0
            signs.append(np.sign(r["delta_t"][0]))
    assert signs[0::2] == signs[1::2]


def test_kdtree_equals_brute_force_with_ties():
    corner, _ = F.generate_corner()
    q = corner + np.float32(0.05)              # grid midpoints: many exact ties
    ib, db = O.nn_brute(q, corner)
    ik, dk = O.kdtree_knn(q, corner, 1)
    assert np.array_equal(ib, ik[:, 0]) and np.array_equal(db, dk[:, 0])
    rng = np.random.default_rng(0)
    t = rng.standard_normal((5000, 3)).astype(np.float32)
    qq = rng.standard_normal((777, 3)).astype(np.float32)
    ib, db = O.knn_brute(qq, t, 20)
    ik, dk = O.kdtree_knn(qq, t, 20, threads=4)
    assert np.array_equal(ib, ik) and np.array_equal(db, dk)


def test_nn_against_scipy():
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(1)
    t = (rng.random((4000, 3)) * 10).astype(np.float32)
    q = (rng.random((500, 3)) * 10).astype(np.float32)
    ib, db = O.nn_brute(q, t)
    d, i = cKDTree(t.astype(np.float64)).query(q.astype(np.float64))
    assert np.mean(ib == i) > 0.999           # float vs double distances may flip near-ties
    assert np.allclose(db, d * d, rtol=1e-5, atol=1e-7)


def test_cov_from_normals_properties():
    rng = np.random.default_rng(2)
    n = rng.standard_normal((200, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    n[0] = [0, 0, 1]; n[1] = [1, 0, 0]; n[2] = [0, 1, 0]; n[3] = [0, 0, 0]
    Cs = O.cov_from_normals(n)
    for k in range(4, 200):
        ev = np.linalg.eigvalsh(Cs[k])
        assert np.allclose(ev, [1e-3, 1, 1], atol=1e-6)
        assert abs(n[k].astype(np.float64) @ Cs[k] @ n[k].astype(np.float64) - 1e-3) < 1e-6
    assert np.allclose(Cs[0], np.diag([1, 1, 1e-3]), atol=1e-12)
    # normal_z below 1e-7 is replaced by 1e-7 without re-normalising (gicp_utils.h:37-40)
    assert np.all(np.isfinite(Cs[1])) and np.all(np.isfinite(Cs[2])) and np.all(np.isfinite(Cs[3]))


def test_cov_knn_plane():
    xyz, _ = F.generate_plane(20, 20)
    Cs = O.cov_knn(xyz, 20, 1e-3)
    assert np.allclose(Cs, np.diag([1, 1, 1e-3]), atol=1e-9)
    assert O.cov_knn(xyz[:10], 20) is None     # gicp.hpp:72-79


def test_cost_gradient_matches_finite_differences():
    src, sn, tgt, tn, T = F.make_pair(400, 3)
    idx, d2 = O.nn_brute(src, tgt)
    M = np.linalg.inv(O.cov_from_normals(sn) + O.cov_from_normals(tn)[idx])
    x = np.array([0.01, -0.02, 0.03, 0.02, -0.01, 0.015])
    si = np.arange(len(src))
    f, g = O.gicp_cost_grad(src, tgt, si, idx, M, x)
    # central differences on a double-precision re-implementation of the same objective
    def cost(xx):
        R = F.rot_zyx(xx[3], xx[4], xx[5])
        r = src.astype(np.float64) @ R.T + xx[:3] - tgt[idx].astype(np.float64)
        return np.einsum("ni,nij,nj->", r, M, r) / len(src)
    gn = np.array([(cost(x + h) - cost(x - h)) / 2e-6 for h in np.eye(6) * 1e-6])
    assert abs(f - cost(x)) < 1e-4 * max(1, abs(f))     # float transform vs double
    assert np.allclose(g, gn, rtol=2e-3, atol=2e-3)


def test_bfgs_and_gn_reach_the_same_fixed_point():
    """The reference's BFGS leaves an alignment when it stops making progress inside the rounding
    noise of its float objective (gicp.hpp:307-310, 263-278), so its end point is defined only up
    to that noise; Gauss-Newton lands on the minimiser of the same objective.  On room-like
    scenes the two end within 1e-4 rad / 2e-4 m of each other, and the reference's own answer
    moves by as much when its initial guess moves by one micrometre, while the Gauss-Newton
    answer does not."""
    moved = np.eye(4, dtype=np.float32)
    moved[0, 3] = 1e-6
    gaps, self_bf, self_gn = [], [], []
    for n, seed, rd, tr in [(3000, 5, 2.0, 0.15), (3000, 6, 3.0, 0.3), (6000, 8, 5.0, 0.5),
                            (8000, 6, 3.0, 0.3)]:
        src, sn, tgt, tn, T = F.make_pair(n, seed, rot_deg=rd, trans=tr)
        out, out_m = {}, {}
        for solver in (O.SOLVER_BFGS, O.SOLVER_GN):
            r = O.gicp_align(O.base_gicp_params(solver=solver), src, sn, tgt, tn, np.eye(4))
            assert r["converged"] and r["delta"] < 1 and r["iterations"] < 50
            out[solver] = r["T"]
            out_m[solver] = O.gicp_align(O.base_gicp_params(solver=solver), src, sn, tgt, tn, moved)["T"]
        gaps.append(O.pose_gap(out[O.SOLVER_BFGS], out[O.SOLVER_GN]))
        self_bf.append(O.pose_gap(out[O.SOLVER_BFGS], out_m[O.SOLVER_BFGS]))
        self_gn.append(O.pose_gap(out[O.SOLVER_GN], out_m[O.SOLVER_GN]))
        assert O.pose_gap(T, out[O.SOLVER_GN])[1] < 0.02
        # stationarity of the reference's objective at both answers
        P = O.base_gicp_params()
        g_bf = np.linalg.norm(O.gicp_objective_at(P, src, sn, tgt, tn, out[O.SOLVER_BFGS])[1])
        g_gn = np.linalg.norm(O.gicp_objective_at(P, src, sn, tgt, tn, out[O.SOLVER_GN])[1])
        assert g_gn < 1e-2 and g_gn < g_bf
    assert max(g[0] for g in gaps) < 1e-4 and max(g[1] for g in gaps) < 2e-4, gaps
    assert max(g[1] for g in self_gn) < 2e-5, self_gn
    assert max(g[1] for g in self_bf) > 5.0 * max(g[1] for g in self_gn), (self_bf, self_gn)


def test_small_algebra():
    rng = np.random.default_rng(4)
    A = rng.standard_normal((3, 3)) + 3 * np.eye(3)
    assert np.allclose(O.mat3_inverse(A), np.linalg.inv(A))
    B = rng.standard_normal((6, 6)); B = B @ B.T + np.eye(6)
    assert np.allclose(O.mat6_inverse(B), np.linalg.inv(B))
    ev, V = O.sym_eig(B)
    assert np.allclose(ev, np.linalg.eigvalsh(B)) and np.allclose(V @ np.diag(ev) @ V.T, B)
    L, D, tr = O.ldlt(B)
    P = np.eye(6)
    for k, t in enumerate(tr):
        P[[k, t]] = P[[t, k]]
    assert np.allclose(L @ np.diag(D) @ L.T, P @ B @ P.T)


def test_pose_algebra():
    from scipy.spatial.transform import Rotation as Rsc
    rng = np.random.default_rng(5)
    q = rng.standard_normal(4); q /= np.linalg.norm(q)
    p = O.pose_from_quat(q, [1, 2, 3])
    R, t = O.pose_to_np(p)
    assert np.allclose(R, Rsc.from_quat([q[1], q[2], q[3], q[0]]).as_matrix())
    xi = O.pose_logmap(p)
    assert np.allclose(xi[:3], Rsc.from_matrix(R).as_rotvec())
    # exp(log) round trip for the translation part via the SE(3) series
    from scipy.linalg import expm
    W = np.array([[0, -xi[2], xi[1]], [xi[2], 0, -xi[0]], [-xi[1], xi[0], 0]])
    X = np.zeros((4, 4)); X[:3, :3] = W; X[:3, 3] = xi[3:]
    E = expm(X)
    assert np.allclose(E[:3, :3], R) and np.allclose(E[:3, 3], t)
    inv = O.pose_inverse(p)
    Ri, ti = O.pose_to_np(inv)
    assert np.allclose(Ri, R.T) and np.allclose(ti, -R.T @ t)


def test_fpfh_oracle_properties():
    # lamp_utils/test/test_point_cloud_utils.cc:153-170 pins the count only (3 descriptors)
    corner, cn = F.generate_corner()
    kp = corner[[33, 133, 233]]
    f = O.fpfh(corner, cn, kp, 1.0)
    assert f.shape == (3, 33)
    assert np.allclose(f.reshape(3, 3, 11).sum(-1), 100, atol=1e-3)
    # descriptors are invariant under a rigid motion of cloud + normals + keypoints (checked on
    # a noisy scan: on the grid fixture angles and radii sit exactly on bin / radius edges)
    xyz, nrm = F.random_room_scan(1500, 3)
    kp = xyz[::300]
    f = O.fpfh(xyz, nrm, kp, 2.0)
    R = F.rot_zyx(0.3, -0.2, 0.7)
    x2 = (xyz.astype(np.float64) @ R.T + [1, 2, 3]).astype(np.float32)
    n2 = (nrm.astype(np.float64) @ R.T).astype(np.float32)
    f2 = O.fpfh(x2, n2, x2[::300], 2.0)
    # axis-aligned room normals put f1 exactly on +-pi, whose sign flips under rounding
    assert np.abs(f - f2).max() < 6.0 and np.median(np.abs(f - f2)) < 1e-3


# ------------------------------------------------------------------ Harris keypoints, SAC-IA
def test_harris_corner_keypoints():
    """lamp_utils::ComputeKeypoints (PointCloudUtils.cc:71-96).  On the three-plane corner the
    only points whose neighbourhood holds all three normals equally are the three copies of the
    origin; their response is 0.04 + det(I/3) - 0.04 = 1/27."""
    corner, cn = F.generate_corner()
    idx, resp = O.harris_keypoints(corner, cn, 1.0)
    assert list(idx) == [0, 100, 200]
    assert np.all(corner[idx] == 0)
    assert np.allclose(resp, 1 / 27, rtol=1e-5)
    # a single plane has a rank-1 normal covariance: response 0 everywhere, no keypoint
    plane, pn = F.generate_plane()
    assert np.all(O.harris_response(plane, pn, 1.0) == 0)
    assert len(O.harris_keypoints(plane, pn, 1.0)[0]) == 0


def test_sac_ia_reference_corner_case():
    """TestSacInitialAlign (test_loop_computation.cc:78-104): corner vs the same corner moved by
    (1, -0.001, 0) is expected to give exactly that translation, identity rotation.  All three
    keypoints sit on the origin, so every sample triple is degenerate and the estimate is the
    centroid shift."""
    corner, cn = F.generate_corner()
    moved, mn, _ = F.shifted_corner()
    T, fit, (i, j) = O.sac_initial_alignment(corner, cn, moved, mn, iterations=500)
    assert len(i) == 3 and len(j) == 3
    assert np.array_equal(T[:3, :3], np.eye(3, dtype=np.float32))
    assert np.allclose(T[:3, 3], [1.0, -0.001, 0.0], atol=1e-7)
    assert fit == 0.0


def test_sac_ia_too_few_keypoints():
    plane, pn = F.generate_plane()
    T, fit, _ = O.sac_initial_alignment(plane, pn, plane, pn, iterations=10)
    assert np.array_equal(T, np.eye(4, dtype=np.float32)) and fit > 1e300


# ------------------------------------------------------------------ TEASER++ restatement
def _teaser_scene(n=200, n_in=60, seed=0, noise=0.05):
    from scipy.spatial.transform import Rotation as Rsc
    rng = np.random.default_rng(seed)
    src = rng.uniform(-10, 10, (n, 3))
    R = Rsc.from_rotvec([0.5, -0.8, 1.2]).as_matrix()
    t = np.array([3.0, -2.0, 1.0])
    dst = src @ R.T + t + rng.normal(0, noise, (n, 3))
    outl = rng.permutation(n)[: n - n_in]
    dst[outl] = rng.uniform(-10, 10, (n - n_in, 3))
    return src, dst, R, t, np.setdiff1d(np.arange(n), outl)


def test_teaser_solver_rejects_70_percent_outliers():
    """Restated TEASER++ (parity unpinned: the library is not in the reference tree and the
    reference's own test of this path is commented out, test_loop_computation.cc:101-123).
    Checked through what the algorithm guarantees: the clique holds inliers only, rotation and
    translation are recovered to the noise level, the truncated error beats the identity's."""
    src, dst, R, t, inl = _teaser_scene()
    r = O.teaser_solve(src, dst)
    assert r["valid"] and set(r["clique"]) <= set(inl) and len(r["clique"]) >= 0.9 * len(inl)
    assert np.abs(r["R"] - R).max() < 5e-3 and np.abs(r["t"] - t).max() < 0.05
    assert np.allclose(r["R"] @ r["R"].T, np.eye(3), atol=1e-12) and np.linalg.det(r["R"]) > 0
    assert O.teaser_error_metric(src, dst, r["R"], r["t"]) < \
        O.teaser_error_metric(src, dst, np.eye(3), np.zeros(3))


def test_teaser_parts():
    from scipy.spatial.transform import Rotation as Rsc
    rng = np.random.default_rng(1)
    R = Rsc.from_rotvec([-1.0, 0.3, 2.0]).as_matrix()
    x = rng.normal(size=(50, 3))
    assert np.abs(O.teaser_gnc_rotation(x, x @ R.T) - R).max() < 1e-12
    # planar and collinear inputs (rank-deficient correlation) still give a proper rotation
    xp = x.copy(); xp[:, 2] = 0
    Rp = O.teaser_gnc_rotation(xp, xp @ R.T)
    assert np.allclose(Rp @ xp.T, R @ xp.T, atol=1e-9) and abs(np.linalg.det(Rp) - 1) < 1e-12
    # scalar TLS: the consensus of {0.9, 1.0, 1.1} wins over two far outliers
    assert abs(O.teaser_tls_scalar([1.0, 1.1, 0.9, 5.0, -3.0], 0.5) - 1.0) < 1e-12
    # clique of a graph made of a 5-clique plus a pendant triangle
    adj = np.zeros((8, 8), np.uint8)
    for a in range(5):
        for b in range(5):
            adj[a + 2, b + 2] = a != b
    adj[0, 1] = adj[1, 0] = adj[0, 2] = adj[2, 0] = adj[1, 2] = adj[2, 1] = 1
    assert list(O.teaser_clique(adj)) == [2, 3, 4, 5, 6]
    assert len(O.teaser_clique(np.zeros((4, 4), np.uint8))) <= 1
    # matches: k nearest target descriptors per source descriptor, ascending distance
    sf = np.zeros((2, 33), np.float32); tf = np.zeros((7, 33), np.float32)
    tf[:, 0] = [5, 1, 3, 0.5, 9, 2, 4]; sf[1, 0] = 9
    m = O.teaser_matches(sf, tf, 5)
    assert m[:5].tolist() == [[0, 3], [0, 1], [0, 5], [0, 2], [0, 6]]
    assert m[5:].tolist() == [[1, 4], [1, 0], [1, 6], [1, 2], [1, 5]]


def test_teaser_initial_alignment_corner():
    """The reference's (commented-out) corner case has 3 keypoints per cloud: 3 x 3 = 9 matches,
    not more than 10, so the guess comes back unchanged (IcpLoopComputation.cc:1002,1078)."""
    corner, cn = F.generate_corner()
    moved, mn, _ = F.shifted_corner()
    g = np.eye(4, dtype=np.float32); g[0, 3] = 0.9
    T, used, n_matches, n_clique = O.teaser_initial_alignment(corner, cn, moved, mn, guess=g)
    assert n_matches == 9 and not used and np.array_equal(T, g)


# ------------------------------------------------------------------ normal estimation
def test_compute_normals_reference_plane():
    """TestPointCloudUtils.ComputeNormals / ExtractNormals (test_point_cloud_utils.cc:39-68):
    every normal of the 100-point plane is (0, 0, 1) within 1e-5."""
    plane, pn = F.generate_plane()
    nrm, cur = O.compute_normals(plane, 10)
    assert nrm.shape == (100, 3)
    assert np.allclose(nrm, [0, 0, 1], atol=1e-5) and np.all(cur == 0)
    assert np.allclose(O.extract_normals(plane, pn), [0, 0, 1], atol=1e-5)
    # a cloud that arrives without normals gets them estimated
    assert np.allclose(O.extract_normals(plane, np.zeros_like(pn)), [0, 0, 1], atol=1e-5)


def test_compute_normals_room():
    xyz, true_n = F.random_room_scan(4000, 11)
    nrm, cur = O.compute_normals(xyz, 10)
    assert np.allclose(np.linalg.norm(nrm, axis=1), 1, atol=1e-5)
    assert np.all((nrm * -xyz).sum(1) >= -1e-6)              # flipped towards the origin
    assert np.median(np.abs((nrm * true_n).sum(1))) > 0.99
    assert np.all((cur >= 0) & (cur <= 1 / 3 + 1e-5))


def test_harris_keypoints_without_normals_reference_corner():
    """lamp_utils/test/test_point_cloud_utils.cc:114-133 pins three keypoints on the corner for the
    ComputeKeypoints overload without normals: (0.3,0.3,0), (0.3,0,0.3), (0,0.3,0.3), the three
    images of one point under the corner's symmetry.  Their responses are equal in exact
    arithmetic and all three survive PCL's non-maxima suppression only because PCL's float sums
    come out bit-identical for them; that depends on the order in which FLANN's radius search
    hands back equidistant grid neighbours, which the reference tree does not determine.  With
    neighbours summed in ascending index the three responses agree to 3e-8 and are the three
    largest of the cloud; the suppression then keeps the one(s) ahead in the last bits."""
    corner, _ = F.generate_corner()
    nrm, _ = O.compute_normals_radius(corner, 1.0)
    resp = O.harris_response(corner, nrm, 1.0)
    top3 = np.argsort(-resp)[:3]
    got = {tuple(np.round(corner[i].astype(np.float64), 5).tolist()) for i in top3}
    assert got == {(0.3, 0.3, 0.0), (0.3, 0.0, 0.3), (0.0, 0.3, 0.3)}
    assert resp[top3].max() - resp[top3].min() < 3e-8 and resp[top3].min() > 5e-9
    assert np.sort(resp)[-4] < resp[top3].min() - 1e-4
    idx, _ = O.harris_keypoints(corner, None, 1.0)
    assert 1 <= len(idx) <= 3 and set(idx.tolist()) <= set(top3.tolist())
    # a plane has no corner response at all
    plane, _ = F.generate_plane()
    assert len(O.harris_keypoints(plane, None, 1.0)[0]) == 0
