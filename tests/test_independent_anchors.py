"""An implementation of the GICP outer loop that shares nothing with the oracle's solvers: numpy for
the objective of gicp.hpp:290-402 in double precision, MINPACK's Levenberg-Marquardt (scipy) for the
inner minimisation.  It anchors what the default back end claims: the Gauss-Newton solver lands on
the minimiser of the reference's objective, the reference's BFGS within its own noise of it."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import fixtures as F

scipy_optimize = pytest.importorskip("scipy.optimize")


def _state_of(T):
    """The six parameters the reference's optimiser starts from (gicp.hpp:235-246)."""
    T = np.asarray(T, np.float64)
    return np.array([T[0, 3], T[1, 3], T[2, 3], np.arctan2(T[2, 1], T[2, 2]), np.arcsin(-T[2, 0]),
                     np.arctan2(T[1, 0], T[0, 0])])


def _gicp_numpy(src, sn, tgt, tn, corr_dist=0.5, iterations=40):
    C1, C2 = O.cov_from_normals(sn), O.cov_from_normals(tn)
    p = src.astype(np.float64)
    T = np.eye(4, dtype=np.float32)
    seen = []
    for _ in range(iterations):
        idx, d2 = O.nn_brute(O.transform_points_f32(T, src), tgt)
        keep = d2 < np.float32(corr_dist * corr_dist)                   # gicp.hpp:476
        R = T[:3, :3].astype(np.float64)
        M = np.linalg.inv(C2[idx[keep]] + R @ C1[keep] @ R.T)           # gicp.hpp:483-497
        L = np.linalg.cholesky(M)                                       # r^T M r = |L^T r|^2
        pk, qk = p[keep], tgt[idx[keep]].astype(np.float64)

        def residuals(x):
            r = pk @ F.rot_zyx(x[3], x[4], x[5]).T + x[:3] - qk
            return np.einsum("nji,nj->ni", L, r).ravel()

        sol = scipy_optimize.least_squares(residuals, _state_of(T), method="lm", xtol=1e-15, ftol=1e-15,
                                           gtol=1e-15, max_nfev=400)
        T_new = O.apply_state(sol.x)                                    # gicp.hpp:619-634, float
        if any(np.array_equal(T_new, s) for s in seen):                 # settled, or a short cycle
            return T_new
        seen.append(T_new)
        T = T_new
    return T


@pytest.mark.parametrize("n,seed,rot_deg,trans", [(2500, 5, 2.0, 0.15), (3000, 8, 4.0, 0.4), (6000, 6, 3.0, 0.3)])
def test_gauss_newton_lands_where_an_independent_minimiser_lands(n, seed, rot_deg, trans):
    src, sn, tgt, tn, _ = F.make_pair(n, seed, rot_deg=rot_deg, trans=trans)
    T_np = _gicp_numpy(src, sn, tgt, tn)
    gn = O.gicp_align(O.base_gicp_params(solver=O.SOLVER_GN), src, sn, tgt, tn, np.eye(4))
    bf = O.gicp_align(O.base_gicp_params(solver=O.SOLVER_BFGS), src, sn, tgt, tn, np.eye(4))
    assert gn["converged"] and bf["converged"]
    # the same float matrix up to its last bit (measured: bit-identical, or one translation entry
    # one float apart: 5e-10 m)
    ang, te = O.pose_gap(T_np, gn["T"])
    assert ang < 1e-8 and te < 1e-8, (ang, te)
    assert np.max(np.abs(T_np.astype(np.float64) - gn["T"].astype(np.float64))) < 2e-7
    # the reference's solver stops inside the noise of its float objective (measured: 5e-5 m,
    # 4e-6 rad away): north_star's 1e-4, and farther from the minimiser than Gauss-Newton
    ang_b, te_b = O.pose_gap(T_np, bf["T"])
    assert ang_b < 1e-4 and te_b < 2e-4, (ang_b, te_b)
    assert te <= te_b and ang <= ang_b


def test_knn_covariances_against_lapack():
    """computeCovariances' k-NN branch (gicp.hpp:83-155) with LAPACK's SVD (numpy) in place of the
    oracle's own Jacobi iteration: 20 nearest neighbours, raw second moments minus the mean's outer
    product, singular values replaced by (1, 1, 1e-3).  A point whose two largest singular values
    nearly coincide has no unique U; the rebuilt matrix does not depend on that (it is
    1 - (1 - 1e-3) u3 u3^T), only a nearly vanishing gap between the two smallest does."""
    xyz, _ = F.random_room_scan(3000, 21)
    got = O.cov_knn(xyz, 20, 1e-3, threads=4)
    idx, _ = O.knn_brute(xyz, xyz, 20)
    P32 = xyz[idx]                                           # n x 20 x 3
    mean = P32.astype(np.float64).mean(axis=1)
    # (the reference multiplies two floats before it widens: `cov(0, 0) += pt.x * pt.x`)
    prod = (P32[:, :, :, None] * P32[:, :, None, :]).astype(np.float64)
    raw = prod.sum(axis=1) / 20.0 - np.einsum("ni,nj->nij", mean, mean)
    U, sv, _ = np.linalg.svd(raw)
    want = np.einsum("nik,k,njk->nij", U, np.array([1.0, 1.0, 1e-3]), U)
    err = np.abs(got - want).max(axis=(1, 2))
    gap = (sv[:, 1] - sv[:, 2]) / sv[:, 0]                   # what u3 is conditioned by
    well = gap > 1e-6
    assert well.mean() > 0.99
    assert err[well].max() < 1e-12, err[well].max()          # measured: 8e-15
    assert np.median(err) < 1e-14


def test_nearest_neighbours_against_scipy_kdtree():
    """The oracle's kd-tree and its exhaustive search against scipy's cKDTree (double precision):
    same neighbours wherever the float distances leave no tie (FLANN's L2_Simple arithmetic differs
    from the double distance in the last float bit at most)."""
    spatial = pytest.importorskip("scipy.spatial")
    rng = np.random.default_rng(3)
    tgt, _ = F.random_room_scan(4000, 31)
    q = tgt[rng.choice(len(tgt), 1500, replace=False)] + rng.normal(0, 0.05, (1500, 3)).astype(np.float32)
    tree = spatial.cKDTree(tgt.astype(np.float64))
    d_ref, i_ref = tree.query(q.astype(np.float64), k=21)
    for k in (1, 20):
        i_kd, d_kd = O.kdtree_knn(q, tgt, k, threads=4)
        i_bf, d_bf = (O.nn_brute(q, tgt) if k == 1 else O.knn_brute(q, tgt, k))
        i_bf, d_bf = np.reshape(i_bf, (len(q), k)), np.reshape(d_bf, (len(q), k))
        assert np.array_equal(d_kd, d_bf)                      # the two oracles: the same floats
        # no tie between a neighbour and the next one (relative gap of the squared distances)
        gaps = np.diff(d_ref[:, :k + 1] ** 2, axis=1) / np.maximum(d_ref[:, 1:k + 1] ** 2, 1e-12)
        clear = gaps.min(axis=1) > 1e-5
        assert clear.mean() > 0.95
        assert np.array_equal(i_kd[clear], i_ref[clear, :k]) and np.array_equal(i_bf[clear], i_ref[clear, :k])
        assert np.allclose(d_kd[clear], d_ref[clear, :k] ** 2, rtol=2e-6, atol=1e-9)


def test_pose_logmap_against_the_matrix_logarithm():
    """gtsam::Pose3::Logmap (restated in the oracle for the transform gate,
    IcpLoopComputation.cc:604-614) against scipy's general matrix logarithm of the 4 x 4 matrix."""
    linalg = pytest.importorskip("scipy.linalg")
    rng = np.random.default_rng(9)
    for _ in range(20):
        w = rng.normal(0, 0.6, 3)
        th = np.linalg.norm(w)
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
        t = rng.normal(0, 2.0, 3)
        # quaternion of R (w, x, y, z)
        qw = 0.5 * np.sqrt(1 + np.trace(R))
        q = np.array([qw, (R[2, 1] - R[1, 2]) / (4 * qw), (R[0, 2] - R[2, 0]) / (4 * qw), (R[1, 0] - R[0, 1]) / (4 * qw)])
        xi = O.pose_logmap(O.pose_from_quat(q, t))
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, t
        Lg = np.real(linalg.logm(T))
        assert np.allclose(xi[:3], [Lg[2, 1], Lg[0, 2], Lg[1, 0]], atol=1e-9)
        assert np.allclose(xi[3:], Lg[:3, 3], atol=1e-9)


def test_knn_normals_against_numpy_eigenvectors():
    """lamp_utils::ComputeNormals (pcl::NormalEstimation over the 10 nearest neighbours): the normal
    is the eigenvector of the neighbourhood covariance with the smallest eigenvalue, up to its
    sign.  LAPACK's symmetric eigensolver on the same neighbourhoods."""
    xyz, _ = F.random_room_scan(3000, 41)
    nrm, _ = O.compute_normals(xyz, 10)
    idx, _ = O.knn_brute(xyz, xyz, 10)
    P = xyz.astype(np.float64)[idx]
    c = P - P.mean(axis=1, keepdims=True)
    ev, V = np.linalg.eigh(np.einsum("nki,nkj->nij", c, c))
    want = V[:, :, 0]
    well = (ev[:, 1] - ev[:, 0]) / np.maximum(ev[:, 2], 1e-30) > 1e-3
    cos = np.abs(np.einsum("ni,ni->n", nrm.astype(np.float64), want))
    assert well.mean() > 0.9
    assert cos[well].min() > 1 - 1e-4, cos[well].min()


def _fpfh_numpy(surf, nrm, keys, radius):
    """FPFH after Rusu et al. (ICRA 2009) as PCL's FPFHEstimation computes it, written from the
    publication's formulas in numpy double precision: SPFH of every surface point within the radius
    of a keypoint (Darboux-frame angles of the point and each neighbour, 11 bins each, increment
    100 / (k - 1)), then the 1/d^2-weighted sum at the keypoint, each block rescaled to sum 100."""
    surf64, nrm64 = surf.astype(np.float64), nrm.astype(np.float64)
    r2 = radius * radius

    def neighbours(p):
        d2 = ((surf64 - p) ** 2).sum(axis=1)
        return np.flatnonzero(d2 < r2), d2

    def bin_of(v, lo, hi):
        return int(min(10, max(0, np.floor(11 * (v - lo) / (hi - lo)))))

    spfh = {}

    def spfh_of(i):
        if i in spfh:
            return spfh[i]
        nb, _ = neighbours(surf64[i])
        h = np.zeros(33)
        for j in nb:
            if j == i:
                continue
            dp = surf64[j] - surf64[i]
            f4 = np.linalg.norm(dp)
            if f4 == 0:
                continue
            n1, n2 = nrm64[i], nrm64[j]
            a1, a2 = n1 @ dp / f4, n2 @ dp / f4
            if np.arccos(abs(a1)) > np.arccos(abs(a2)):      # the frame sits on the point whose normal
                n1, n2, dp, f3 = n2, n1, -dp, -a2            # is closer to the connecting line
            else:
                f3 = a1
            v = np.cross(dp, n1)
            vn = np.linalg.norm(v)
            if vn == 0:
                continue
            v /= vn
            w = np.cross(n1, v)
            f2 = v @ n2
            f1 = np.arctan2(w @ n2, n1 @ n2)
            h[bin_of(f1, -np.pi, np.pi)] += 1
            h[11 + bin_of(f2, -1, 1)] += 1
            h[22 + bin_of(f3, -1, 1)] += 1
        spfh[i] = h * (100.0 / (len(nb) - 1)) if len(nb) > 1 else h
        return spfh[i]

    out = np.zeros((len(keys), 33))
    for q, kp in enumerate(keys.astype(np.float64)):
        nb, d2 = neighbours(kp)
        acc = np.zeros(33)
        for j in nb:
            if d2[j] == 0:
                continue
            acc += spfh_of(j) / d2[j]
        for b in range(3):
            tot = acc[11 * b:11 * b + 11].sum()
            if tot != 0:
                out[q, 11 * b:11 * b + 11] = acc[11 * b:11 * b + 11] * (100.0 / tot)
    return out


def test_fpfh_against_a_numpy_restatement():
    """The oracle's FPFH (C, float arithmetic where PCL's is float) against the formulas written
    out in numpy in double: the same histograms to float precision (measured: 2e-5 at most).  The
    room's axis-aligned normals are tilted a little first: with them many angles sit exactly on a
    bin edge, where float and double fall on different sides (0.9 % of the entries then differ, by
    up to 0.5 of 100)."""
    xyz, nrm = F.random_room_scan(700, 11, noise=0.02)
    rng = np.random.default_rng(11)
    nrm = nrm + rng.normal(0, 0.05, nrm.shape).astype(np.float32)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    keys = xyz[::100]
    got = O.fpfh(xyz, nrm, keys, 1.2)
    want = _fpfh_numpy(xyz, nrm, keys, 1.2)
    assert got.shape == want.shape == (7, 33)
    assert np.allclose(want.reshape(7, 3, 11).sum(-1), 100, atol=1e-9)
    err = np.abs(got - want)
    assert np.median(err) < 1e-6 and err.max() < 1e-3, (np.median(err), err.max())


def test_harris_response_against_numpy():
    """pcl::HarrisKeypoint3D's HARRIS response on given normals (PointCloudUtils.cc:78-96): mean of
    n n^T over the neighbours within the radius, r = 0.04 + det - 0.04 trace^2 in float.  The same
    in numpy double."""
    xyz, nrm = F.random_room_scan(1200, 13, noise=0.02)
    got = O.harris_response(xyz, nrm, 0.6)
    p, n = xyz.astype(np.float64), nrm.astype(np.float64)
    d2 = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1)
    inside = d2 < 0.36
    want = np.zeros(len(p))
    for i in range(len(p)):
        nb = n[inside[i]]
        C = nb.T @ nb / len(nb)
        tr = np.trace(C)
        want[i] = 0.04 + np.linalg.det(C) - 0.04 * tr * tr if tr != 0 else 0.0
    # (det and 0.04 trace^2 cancel to ~1e-3 of their size: float sums leave ~1e-6 absolute)
    assert np.allclose(got, want, atol=5e-6), np.abs(got - want).max()
    assert np.abs(want).max() > 1e-3


def test_teaser_scalar_tls_against_exhaustive_search():
    """The scalar TLS estimator of TEASER++ (adaptive voting: one sweep over the 2n interval ends
    with running sums) against a direct evaluation of what the sweep enumerates: for a point in
    every gap between two interval ends, the measurements whose interval [a_i - c, a_i + c] holds
    it, their mean, and the library's cost for that consensus set -- squared residuals of the
    members plus c (the library adds the ranges, not their squares) per measurement left out."""
    rng = np.random.default_rng(17)
    for trial in range(40):
        n = int(rng.integers(5, 40))
        a = np.concatenate([rng.normal(2.0, 0.15, n), rng.uniform(-8, 8, int(rng.integers(0, n)))])
        c = float(rng.uniform(0.2, 0.8))
        got = O.teaser_tls_scalar(a, c)
        ends = np.sort(np.concatenate([a - c, a + c]))
        best, best_x = np.inf, None
        for p in 0.5 * (ends[:-1] + ends[1:]):
            member = np.abs(a - p) <= c
            if not member.any():
                continue
            x = a[member].mean()
            cost = ((x - a[member]) ** 2).sum() + c * (~member).sum()
            if cost < best:
                best, best_x = cost, x
        assert abs(got - best_x) < 1e-9, (trial, got, best_x)


def test_teaser_gnc_rotation_is_the_least_squares_rotation_of_its_inliers():
    """GNC-TLS ends with weights 0 or 1: with no measurement beyond the noise bound the rotation it
    returns is the plain least-squares rotation of all of them (Kabsch / Horn through LAPACK's
    SVD); with gross outliers added, that of the inliers alone."""
    from scipy.spatial.transform import Rotation as Rsc
    rng = np.random.default_rng(23)

    def kabsch(x, y):
        U, _, Vt = np.linalg.svd(y.T @ x)
        D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
        return U @ D @ Vt

    R = Rsc.from_rotvec([0.4, -0.9, 0.3]).as_matrix()
    x = rng.normal(0, 3.0, (80, 3))
    y = x @ R.T + rng.normal(0, 0.02, x.shape)
    got = O.teaser_gnc_rotation(x, y, noise_bound=0.5)
    assert np.abs(got - kabsch(x, y)).max() < 1e-6
    y_out = y.copy()
    y_out[:25] = rng.normal(0, 6.0, (25, 3))                 # a third of the measurements are junk
    got = O.teaser_gnc_rotation(x, y_out, noise_bound=0.5)
    # (an outlier may land within the bound of its prediction by chance: compare on those that do not)
    far = np.linalg.norm(y_out[:25] - x[:25] @ R.T, axis=1) > 1.0
    keep = np.r_[np.flatnonzero(~far), np.arange(25, 80)]
    assert np.abs(got - kabsch(x[keep], y_out[keep])).max() < 1e-6


def test_teaser_clique_against_exact_search():
    """The greedy clique heuristic (what TEASER++ runs by default through PMC) on random graphs with
    a planted clique, against an exact maximum-clique search (Bron-Kerbosch with pivoting, written
    here): what it returns is a clique, and on these graphs -- sparse noise around one dense
    consistent set, the shape of a TEASER++ consistency graph -- it is a maximum one."""
    rng = np.random.default_rng(29)

    def max_clique(adj):
        n = len(adj)
        nbrs = [set(np.flatnonzero(adj[i])) for i in range(n)]
        best = []

        def expand(R, P, X):
            nonlocal best
            if not P and not X:
                if len(R) > len(best):
                    best = list(R)
                return
            if len(R) + len(P) <= len(best):
                return
            pivot = max(P | X, key=lambda u: len(P & nbrs[u]))
            for v in list(P - nbrs[pivot]):
                expand(R + [v], P & nbrs[v], X & nbrs[v])
                P = P - {v}
                X = X | {v}

        expand([], set(range(n)), set())
        return best

    exact_hits = 0
    for trial in range(12):
        n, k = 60, 14
        adj = (rng.random((n, n)) < 0.15).astype(np.uint8)
        adj = np.triu(adj, 1)
        adj = adj + adj.T
        planted = rng.choice(n, k, replace=False)
        adj[np.ix_(planted, planted)] = 1
        np.fill_diagonal(adj, 0)
        got = list(O.teaser_clique(adj))
        assert all(adj[a, b] for i, a in enumerate(got) for b in got[i + 1:])     # a clique
        assert len(got) >= k
        exact_hits += len(got) == len(max_clique(adj))
    assert exact_hits == 12


def test_fitness_score_against_scipy():
    """pcl getFitnessScore (mean squared distance of the aligned source to its nearest target,
    IcpLoopComputation.cc:586) with scipy's cKDTree and double arithmetic."""
    spatial = pytest.importorskip("scipy.spatial")
    src, _, tgt, _, T = F.make_pair(3000, 51, rot_deg=3.0, trans=0.3)
    T = np.asarray(T, np.float32)
    got, idx = O.fitness_score(src, tgt, T)
    moved = src.astype(np.float64) @ T[:3, :3].astype(np.float64).T + T[:3, 3].astype(np.float64)
    d, i = spatial.cKDTree(tgt.astype(np.float64)).query(moved)
    assert abs(got - np.mean(d ** 2)) <= 2e-6 * np.mean(d ** 2)
    assert (idx == i).mean() > 0.999


def test_voxel_grid_against_numpy():
    """pcl::VoxelGrid with a cubic leaf (LampPcldFilter.cc:12-72): one centroid per occupied voxel,
    voxels counted from floor(min / leaf).  The same grouping with numpy; PCL's order of the output
    (by voxel index) is not part of the comparison."""
    xyz, _ = F.random_room_scan(5000, 61, noise=0.01)
    leaf = 0.25
    got = O.voxel_grid(xyz, leaf)
    cell = np.floor(xyz.astype(np.float64) / leaf).astype(np.int64)
    _, inv, cnt = np.unique(cell, axis=0, return_inverse=True, return_counts=True)
    want = np.zeros((len(cnt), 3))
    np.add.at(want, inv.ravel(), xyz.astype(np.float64))
    want /= cnt[:, None]
    assert len(got) == len(want)
    order_g = np.lexsort(np.floor(got.astype(np.float64) / leaf).T)
    order_w = np.lexsort(np.floor(want / leaf).T)
    assert np.allclose(got[order_g], want[order_w], atol=2e-5)
