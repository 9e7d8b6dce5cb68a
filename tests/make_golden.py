"""Regenerates tests/golden/*.npz from the CPU oracle (run: python -m tests.make_golden).

No reference code can run in this environment, so the vectors come from the oracle after it has
been pinned to the reference's own known answers (tests/test_oracle.py); they freeze the oracle's
behaviour so that a later change to either the oracle or the HIP path shows up as a diff."""
import os

import numpy as np

from oracle import oracle as O
from tests import fixtures as F

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def corner_alignment():
    corner, cn = F.generate_corner()
    moved, mn, T = F.shifted_corner()
    guess = np.eye(4, dtype=np.float32)
    guess[:3, 3] = [-0.9, 0.1, -0.1]
    out = {}
    for name, solver in (("bfgs", O.SOLVER_BFGS), ("gn", O.SOLVER_GN)):
        r = O.gicp_align(O.base_gicp_params(solver=solver, use_kdtree=0), moved, mn, corner, cn,
                         guess, trace_cap=8)
        n_it = min(r["iterations"], 8)
        out[f"{name}_T"] = r["T"]
        out[f"{name}_iterations"] = np.int32(r["iterations"])
        out[f"{name}_trace_idx"] = r["trace_idx"][:n_it]
        out[f"{name}_trace_T"] = r["trace_T"][:n_it]
    fit, idx = O.fitness_score(moved, corner, out["gn_T"], use_kdtree=False)
    cov, Ap = O.icp_cov_point_plane(moved, cn, idx, out["gn_T"])
    out.update(fitness=np.float64(fit), corr=idx, covariance=cov, Ap=Ap, guess=guess)
    return out


def plane_ap():
    xyz, nrm = F.generate_plane()
    qn = O.normalize_cloud(xyz)
    return dict(normalized=qn, Ap=O.compute_ap(qn, nrm, np.arange(100), np.eye(4)),
                observability=O.icp_observability(xyz, nrm),
                cov_from_normals=O.cov_from_normals(nrm))


def room_pair():
    src, sn, tgt, tn, T = F.make_pair(2000, 1234, rot_deg=3.0, trans=0.25)
    r = O.gicp_align(O.base_gicp_params(solver=O.SOLVER_GN, use_kdtree=0), src, sn, tgt, tn,
                     np.eye(4), trace_cap=4)
    idx0, d0 = O.nn_brute(src, tgt)
    return dict(T_true=T, T=r["T"], iterations=np.int32(r["iterations"]), nn_idx_iter0=idx0,
                nn_d2_iter0=d0, cov_knn_src_first64=O.cov_knn(src, 20, 1e-3)[:64])


def front_end():
    """Rows around the alignment: normals, keyed-scan filters, Harris / FPFH / SAC-IA / TEASER++
    initialisation on a 3000-point room pair."""
    src, sn, tgt, tn, T = F.make_pair(3000, 4321, rot_deg=3.0, trans=0.3)
    nrm, cur = O.compute_normals(src, 10)
    vox, vox_n = O.voxel_grid(src, 0.35, sn)
    keep = O.random_sample_indices(len(src), 700, 17)
    T_sac, fit_sac, _ = O.sac_initial_alignment(src, sn, tgt, tn, harris_radius=0.3, iterations=60)
    T_te, used, n_matches, n_clique = O.teaser_initial_alignment(src, sn, tgt, tn, harris_radius=0.3)
    return dict(normals=nrm, curvature=cur, voxel_xyz=vox, voxel_normals=vox_n, sample_idx=keep,
                sac_T=T_sac, sac_fitness=np.float64(fit_sac), teaser_T=T_te,
                teaser_used=np.int32(used), teaser_matches=np.int32(n_matches),
                teaser_clique=np.int32(n_clique),
                observability=O.icp_observability(src, sn))


def main():
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "front_end_3000.npz"), **front_end())
    np.savez_compressed(os.path.join(OUT, "corner_alignment.npz"), **corner_alignment())
    np.savez_compressed(os.path.join(OUT, "plane_ap.npz"), **plane_ap())
    np.savez_compressed(os.path.join(OUT, "room_pair_2000.npz"), **room_pair())


if __name__ == "__main__":
    main()
