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
