"""ctypes binding of the CPU oracle (oracle/liblamp_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the lamp_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblamp_oracle.so")


def build(force=False):
    """Compile the oracle with gcc (a few seconds)."""
    srcs = ["lo_linalg.c", "lo_kdtree.c", "lo_gicp.c", "lo_icp_loop.c", "lo_fpfh.c", "lo_features.c", "lamp_oracle.h"]
    if not force and os.path.exists(_LIB_PATH):
        newest = max(os.path.getmtime(os.path.join(_HERE, s)) for s in srcs)
        if os.path.getmtime(_LIB_PATH) >= newest:
            return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liblamp_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


class GicpParams(C.Structure):
    _fields_ = [("corr_dist", C.c_double), ("tf_epsilon", C.c_double),
                ("rotation_epsilon", C.c_double), ("max_iterations", C.c_int),
                ("max_inner_iterations", C.c_int), ("k_correspondences", C.c_int),
                ("gicp_epsilon", C.c_double), ("cov_from_knn", C.c_int), ("solver", C.c_int),
                ("use_kdtree", C.c_int), ("threads", C.c_int), ("gradient_tol", C.c_double),
                ("gn_cycle_exit", C.c_int), ("residual_double", C.c_int), ("sum_order", C.c_int)]


class GicpResult(C.Structure):
    _fields_ = [("final_transformation", C.c_float * 16), ("converged", C.c_int),
                ("iterations", C.c_int), ("last_num_correspondences", C.c_int),
                ("failure", C.c_int), ("last_delta", C.c_double), ("cycle_exit", C.c_int),
                ("pad_", C.c_int)]


class Pose(C.Structure):
    _fields_ = [("R", C.c_double * 9), ("t", C.c_double * 3)]


class IcpParams(C.Structure):
    _fields_ = [("gicp", GicpParams), ("init_method", C.c_int), ("covariance_method", C.c_int),
                ("max_tolerable_fitness", C.c_double), ("transform_thresholding", C.c_int),
                ("max_translation", C.c_double), ("max_rotation_deg", C.c_double),
                ("rot_sigma", C.c_double), ("trans_sigma", C.c_double),
                ("use_fixed_covariances", C.c_int)]


class Alignment(C.Structure):
    _fields_ = [("status", C.c_int), ("T_icp", C.c_float * 16), ("delta", Pose),
                ("fitness", C.c_double), ("covariance", C.c_double * 36), ("iterations", C.c_int),
                ("converged", C.c_int)]


SOLVER_BFGS, SOLVER_GN = 0, 1
INIT_IDENTITY, INIT_ODOMETRY, INIT_ODOM_ROTATION, INIT_FEATURES, INIT_TEASERPP, INIT_CANDIDATE = range(6)
COV_POINT2POINT, COV_POINT2PLANE = 0, 1

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.lo_fitness_score.restype = C.c_double
        _lib.lo_mat6_det.restype = C.c_double
        _lib.lo_kdtree_build.restype = C.c_void_p
        _lib.lo_sac_ia.restype = C.c_double
        _lib.lo_teaser_error_metric.restype = C.c_double
        _lib.lo_teaser_tls_scalar.restype = C.c_double
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def base_gicp_params(**kw):
    """icp_lc block of the reference's base-station config
    (loop_closure/config/laser_parameters.yaml:224-243) + SetupICP (IcpLoopComputation.cc:194-203)."""
    p = GicpParams()
    lib().lo_gicp_default_params(C.byref(p))
    p.corr_dist = 0.5
    p.tf_epsilon = 1e-10
    p.max_iterations = 200
    p.max_inner_iterations = 50
    p.threads = 1
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def transform_points_f32(T, xyz):
    T = _f32(T).reshape(16)
    xyz = _f32(xyz)
    out = np.empty_like(xyz)
    lib().lo_transform_points_f32(_p(T, C.c_float), _p(xyz, C.c_float), len(xyz), _p(out, C.c_float))
    return out


def transform_points_f64(T, xyz):
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    xyz = _f32(xyz)
    out = np.empty_like(xyz)
    lib().lo_transform_points_f64(_p(T, C.c_double), _p(xyz, C.c_float), len(xyz), _p(out, C.c_float))
    return out


def mat4f_mul(a, b):
    a = _f32(a).reshape(16)
    b = _f32(b).reshape(16)
    out = np.empty(16, np.float32)
    lib().lo_mat4f_mul(_p(a, C.c_float), _p(b, C.c_float), _p(out, C.c_float))
    return out.reshape(4, 4)


def nn_brute(q, t):
    q, t = _f32(q), _f32(t)
    idx = np.empty(len(q), np.int32)
    d2 = np.empty(len(q), np.float32)
    lib().lo_nn_brute(_p(q, C.c_float), len(q), _p(t, C.c_float), len(t), _p(idx, C.c_int32),
                      _p(d2, C.c_float))
    return idx, d2


def knn_brute(q, t, k):
    q, t = _f32(q), _f32(t)
    idx = np.empty((len(q), k), np.int32)
    d2 = np.empty((len(q), k), np.float32)
    lib().lo_knn_brute(_p(q, C.c_float), len(q), _p(t, C.c_float), len(t), k, _p(idx, C.c_int32),
                       _p(d2, C.c_float))
    return idx, d2


def kdtree_knn(q, t, k=1, threads=1):
    q, t = _f32(q), _f32(t)
    idx = np.empty((len(q), k), np.int32)
    d2 = np.empty((len(q), k), np.float32)
    L = lib()
    tree = C.c_void_p(L.lo_kdtree_build(_p(t, C.c_float), len(t)))
    L.lo_kdtree_knn(tree, _p(q, C.c_float), len(q), k, _p(idx, C.c_int32), _p(d2, C.c_float), threads)
    L.lo_kdtree_free(tree)
    return idx, d2


def cov_from_normals(normals):
    normals = _f32(normals)
    out = np.empty((len(normals), 3, 3), np.float64)
    lib().lo_cov_from_normals(_p(normals, C.c_float), len(normals), _p(out, C.c_double))
    return out


def gicp_objective_at(params, src_xyz, src_normals, tgt_xyz, tgt_normals, T):
    """Measurement helper: the reference's objective and gradient (6 Euler parameters) at the
    stationary point T claims to be.  Returns (f, g[6], m) or None with fewer than 4 matches."""
    a = [_f32(v) for v in (src_xyz, src_normals, tgt_xyz, tgt_normals)]
    T = _f32(T).reshape(16)
    f = C.c_double()
    g = np.zeros(6, np.float64)
    m = C.c_int()
    rc = lib().lo_gicp_objective_at(C.byref(params), _p(a[0], C.c_float), _p(a[1], C.c_float),
                                    len(a[0]), _p(a[2], C.c_float), _p(a[3], C.c_float), len(a[2]),
                                    _p(T, C.c_float), C.byref(f), _p(g, C.c_double), C.byref(m))
    return None if rc != 0 else (f.value, g, m.value)


def pose_gap(Ta, Tb):
    """(rotation angle in rad, translation norm in m) of Ta^-1 Tb.  The angle comes from the
    antisymmetric part (sin) and the trace (cos) together: arccos of the trace alone cannot
    resolve angles under ~3e-4 rad when the entries are float32."""
    dT = np.linalg.inv(np.asarray(Ta, np.float64)) @ np.asarray(Tb, np.float64)
    R = dT[:3, :3]
    s = 0.5 * np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return float(np.arctan2(s, 0.5 * (np.trace(R) - 1.0))), float(np.linalg.norm(dT[:3, 3]))


def cov_knn(xyz, k=20, eps=1e-3, threads=1):
    xyz = _f32(xyz)
    out = np.empty((len(xyz), 3, 3), np.float64)
    rc = lib().lo_cov_knn(_p(xyz, C.c_float), len(xyz), k, C.c_double(eps), _p(out, C.c_double), threads)
    return out if rc == 0 else None


def gicp_align(params, src_xyz, src_normals, tgt_xyz, tgt_normals, guess=None, trace_cap=0):
    src_xyz, src_normals = _f32(src_xyz), _f32(src_normals)
    tgt_xyz, tgt_normals = _f32(tgt_xyz), _f32(tgt_normals)
    guess = _f32(np.eye(4) if guess is None else guess).reshape(16)
    res = GicpResult()
    tr_idx = tr_T = None
    pi = pT = None
    if trace_cap > 0:
        tr_idx = np.full((trace_cap, len(src_xyz)), -2, np.int32)
        tr_T = np.zeros((trace_cap, 16), np.float32)
        pi, pT = _p(tr_idx, C.c_int32), _p(tr_T, C.c_float)
    lib().lo_gicp_align(C.byref(params), _p(src_xyz, C.c_float), _p(src_normals, C.c_float),
                        len(src_xyz), _p(tgt_xyz, C.c_float), _p(tgt_normals, C.c_float),
                        len(tgt_xyz), _p(guess, C.c_float), C.byref(res), pi, pT, trace_cap)
    out = {
        "T": np.array(res.final_transformation, np.float32).reshape(4, 4),
        "converged": bool(res.converged),
        "iterations": res.iterations,
        "m": res.last_num_correspondences,
        "failure": res.failure,
        "delta": res.last_delta,
        "cycle_exit": bool(res.cycle_exit),
    }
    if trace_cap > 0:
        out["trace_idx"] = tr_idx
        out["trace_T"] = tr_T.reshape(trace_cap, 4, 4)
    return out


def gicp_cost_grad(src_guess, tgt, src_idx, tgt_idx, mahal, x):
    src_guess, tgt = _f32(src_guess), _f32(tgt)
    si = np.ascontiguousarray(src_idx, np.int32)
    ti = np.ascontiguousarray(tgt_idx, np.int32)
    mahal = np.ascontiguousarray(mahal, np.float64)
    x = np.ascontiguousarray(x, np.float64)
    f = C.c_double()
    g = np.empty(6, np.float64)
    lib().lo_gicp_cost_grad(_p(src_guess, C.c_float), _p(tgt, C.c_float), _p(si, C.c_int32),
                            _p(ti, C.c_int32), len(si), _p(mahal, C.c_double), _p(x, C.c_double),
                            C.byref(f), _p(g, C.c_double))
    return f.value, g


def apply_state(x):
    x = np.ascontiguousarray(x, np.float64)
    T = np.empty(16, np.float32)
    lib().lo_apply_state(_p(x, C.c_double), _p(T, C.c_float))
    return T.reshape(4, 4)


def fitness_score(src, tgt, T, use_kdtree=True, threads=1):
    src, tgt = _f32(src), _f32(tgt)
    T = _f32(T).reshape(16)
    idx = np.empty(len(src), np.int32)
    f = lib().lo_fitness_score(_p(src, C.c_float), len(src), _p(tgt, C.c_float), len(tgt),
                               _p(T, C.c_float), _p(idx, C.c_int32), int(use_kdtree), threads)
    return f, idx


def normalize_cloud(xyz):
    xyz = _f32(xyz)
    out = np.empty_like(xyz)
    lib().lo_normalize_cloud(_p(xyz, C.c_float), len(xyz), _p(out, C.c_float))
    return out


def compute_ap(query_normalized, ref_normals, corr, T):
    qn, rn = _f32(query_normalized), _f32(ref_normals)
    corr = np.ascontiguousarray(corr, np.int64)
    T = _f32(T).reshape(16)
    Ap = np.empty(36, np.float64)
    lib().lo_compute_ap(_p(qn, C.c_float), len(qn), _p(rn, C.c_float), len(rn), _p(corr, C.c_int64),
                        len(corr), _p(T, C.c_float), _p(Ap, C.c_double))
    return Ap.reshape(6, 6)


def icp_observability(xyz, normals):
    xyz, normals = _f32(xyz), _f32(normals)
    e = np.empty(3, np.float64)
    lib().lo_icp_observability(_p(xyz, C.c_float), _p(normals, C.c_float), len(xyz), _p(e, C.c_double))
    return e


def icp_cov_point_plane(scan1_xyz, scan2_normals, corr, T, rot_sigma=0.01, trans_sigma=0.01):
    s1, n2 = _f32(scan1_xyz), _f32(scan2_normals)
    corr = np.ascontiguousarray(corr, np.int64)
    T = _f32(T).reshape(16)
    cov = np.empty(36, np.float64)
    Ap = np.empty(36, np.float64)
    lib().lo_icp_cov_point_plane(_p(s1, C.c_float), len(s1), _p(n2, C.c_float), len(n2),
                                 _p(corr, C.c_int64), len(corr), _p(T, C.c_float),
                                 C.c_double(rot_sigma), C.c_double(trans_sigma), _p(cov, C.c_double),
                                 _p(Ap, C.c_double))
    return cov.reshape(6, 6), Ap.reshape(6, 6)


def icp_cov_point_point(aligned_xyz, T, fitness):
    a = _f32(aligned_xyz)
    T = _f32(T).reshape(16)
    cov = np.empty(36, np.float64)
    lib().lo_icp_cov_point_point(_p(a, C.c_float), len(a), _p(T, C.c_float), C.c_double(fitness),
                                 _p(cov, C.c_double))
    return cov.reshape(6, 6)


def fpfh(surface_xyz, surface_normals, keypoints_xyz, radius):
    sx, sn, kx = _f32(surface_xyz), _f32(surface_normals), _f32(keypoints_xyz)
    out = np.empty((len(kx), 33), np.float32)
    lib().lo_fpfh(_p(sx, C.c_float), _p(sn, C.c_float), len(sx), _p(kx, C.c_float), len(kx),
                  C.c_double(radius), _p(out, C.c_float))
    return out


def compute_normals_radius(xyz, radius):
    """pcl::NormalEstimation over a radius search (HarrisKeypoint3D's own normals)."""
    xyz = _f32(xyz)
    nrm = np.empty((len(xyz), 3), np.float32)
    cur = np.empty(len(xyz), np.float32)
    lib().lo_compute_normals_radius(_p(xyz, C.c_float), len(xyz), C.c_double(radius),
                                    _p(nrm, C.c_float), _p(cur, C.c_float))
    return nrm, cur


def harris_keypoints(xyz, normals, radius, threshold=5e-9, nms=True):
    """normals=None: the ComputeKeypoints overload without normals (PointCloudUtils.cc:61-76)."""
    xyz = _f32(xyz)
    if normals is None:
        normals = compute_normals_radius(xyz, radius)[0]
    xyz, normals = _f32(xyz), _f32(normals)
    idx = np.empty(len(xyz), np.int32)
    resp = np.empty(len(xyz), np.float32)
    k = lib().lo_harris_keypoints(_p(xyz, C.c_float), _p(normals, C.c_float), len(xyz),
                                  C.c_double(radius), C.c_double(threshold), int(nms),
                                  _p(idx, C.c_int32), _p(resp, C.c_float))
    return idx[:k].copy(), resp[:k].copy()


def harris_response(xyz, normals, radius):
    xyz, normals = _f32(xyz), _f32(normals)
    resp = np.empty(len(xyz), np.float32)
    lib().lo_harris_response(_p(xyz, C.c_float), _p(normals, C.c_float), len(xyz),
                             C.c_double(radius), _p(resp, C.c_float))
    return resp


def sac_ia(src_kp, src_feat, tgt_kp, tgt_feat, max_iterations, k_corr=5, guess=None, seed=12345):
    sk, sf, tk, tf = _f32(src_kp), _f32(src_feat), _f32(tgt_kp), _f32(tgt_feat)
    g = _f32(np.eye(4) if guess is None else guess).reshape(16)
    T = np.empty(16, np.float32)
    fit = lib().lo_sac_ia(_p(sk, C.c_float), _p(sf, C.c_float), len(sk), _p(tk, C.c_float),
                          _p(tf, C.c_float), len(tk), max_iterations, k_corr, _p(g, C.c_float),
                          C.c_uint32(seed), _p(T, C.c_float))
    return T.reshape(4, 4), fit


def sac_initial_alignment(src, src_n, tgt, tgt_n, harris_radius=0.1, harris_threshold=5e-9,
                          features_radius=2.0, iterations=500, guess=None, seed=0x5AC1A000):
    """GetSacInitialAlignment (IcpLoopComputation.cc:650-705): Harris keypoints, FPFH at the
    keypoints over the full cloud, SAC-IA.  Returns T, fitness, (source, target) keypoint ids."""
    src, tgt = _f32(src), _f32(tgt)
    i, _ = harris_keypoints(src, src_n, harris_radius, harris_threshold)
    j, _ = harris_keypoints(tgt, tgt_n, harris_radius, harris_threshold)
    fs = fpfh(src, src_n, src[i], features_radius)
    ft = fpfh(tgt, tgt_n, tgt[j], features_radius)
    T, fit = sac_ia(src[i], fs, tgt[j], ft, iterations, 5, guess, seed)
    return T, fit, (i, j)


def compute_normals(xyz, k=10, threads=4):
    """lamp_utils::ComputeNormals (k-NN search).  Returns normals [n][3], curvature [n]."""
    xyz = _f32(xyz)
    nrm = np.empty((len(xyz), 3), np.float32)
    cur = np.empty(len(xyz), np.float32)
    lib().lo_compute_normals(_p(xyz, C.c_float), len(xyz), k, _p(nrm, C.c_float),
                             _p(cur, C.c_float), threads)
    return nrm, cur


def extract_normals(xyz, stored, k=10, threads=4):
    xyz, stored = _f32(xyz), _f32(stored)
    nrm = np.empty((len(xyz), 3), np.float32)
    lib().lo_extract_normals(_p(xyz, C.c_float), _p(stored, C.c_float), len(xyz), k,
                             _p(nrm, C.c_float), threads)
    return nrm


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def voxel_grid(xyz, leaf_size, normals=None):
    """pcl::VoxelGrid, cubic leaf.  Returns xyz, or (xyz, normals) when normals are given."""
    x = _f32(xyz).reshape(-1, 3)
    nn = None if normals is None else _f32(normals).reshape(-1, 3)
    out = np.empty((len(x), 3), np.float32)
    out_n = None if nn is None else np.empty((len(x), 3), np.float32)
    f = lib().lo_voxel_grid
    f.restype = C.c_size_t
    m = f(_p(x, C.c_float), None if nn is None else _p(nn, C.c_float), C.c_size_t(len(x)),
          C.c_double(leaf_size), _p(out, C.c_float), None if out_n is None else _p(out_n, C.c_float))
    return out[:m].copy() if nn is None else (out[:m].copy(), out_n[:m].copy())


def random_sample_indices(n, n_keep, seed=0):
    idx = np.empty(max(n, 1), np.int32)
    f = lib().lo_random_sample
    f.restype = C.c_size_t
    m = f(C.c_size_t(n), C.c_size_t(max(int(n_keep), 0)), C.c_uint32(seed & 0xFFFFFFFF), _p(idx, C.c_int32))
    return idx[:m].copy()


class LampPcldFilterOracle:
    """LampPcldFilter (lamp_utils/src/LampPcldFilter.cc:12-72) on the CPU restatements; params is
    any object with the reference's field names."""

    def __init__(self, params, seed=0):
        self.p = params
        self.leaf = (params.adaptive_max_grid + params.adaptive_min_grid) / 2.0
        self.prev_obs = 0.0
        self.seed, self.count = seed, 0

    def filter(self, xyz, normals):
        xyz, normals = self.adaptive_grid_filter(xyz, normals)
        if self.p.random_filter:
            keep = random_sample_indices(len(xyz), int((1.0 - self.p.decimate_percentage) * len(xyz)),
                                         self.seed + self.count)
            xyz, normals = xyz[keep], normals[keep]
        self.count += 1
        return xyz, normals

    def adaptive_grid_filter(self, xyz, normals):
        p = self.p
        xyz, normals = _f32(xyz).reshape(-1, 3), _f32(normals).reshape(-1, 3)
        if len(xyz) < p.adaptive_grid_target:
            return xyz, normals
        xyz, normals = voxel_grid(xyz, self.leaf, normals)
        size_factor = len(xyz) / float(p.adaptive_grid_target)
        obs_factor = 0.0
        if p.observability_check:
            obs = float(np.min(icp_observability(xyz, normals))) / float(len(xyz))
            self.prev_obs = obs                      # the first-cloud flag is never set
            with np.errstate(divide="ignore", invalid="ignore"):
                obs_factor = float((np.float64(self.prev_obs) - obs) / np.float64(self.prev_obs))
        v = self.leaf * (size_factor - abs(size_factor - 1) * obs_factor)
        m = v if p.adaptive_min_grid < v else p.adaptive_min_grid
        self.leaf = m if m < p.adaptive_max_grid else p.adaptive_max_grid
        return xyz, normals


def teaser_matches(src_feat, tgt_feat, k=5):
    sf, tf = _f32(src_feat), _f32(tgt_feat)
    pairs = np.empty((len(sf) * k, 2), np.int32)
    n = lib().lo_teaser_matches(_p(sf, C.c_float), len(sf), _p(tf, C.c_float), len(tf), k,
                                _p(pairs, C.c_int32))
    return pairs[:n].copy()


def teaser_clique(adj):
    adj = np.ascontiguousarray(adj, np.uint8)
    out = np.empty(len(adj), np.int32)
    m = lib().lo_teaser_clique(_p(adj, C.c_uint8), len(adj), _p(out, C.c_int32))
    return out[:m].copy()


def teaser_gnc_rotation(src, dst, noise_bound=0.5, max_iterations=100, gnc_factor=1.4,
                        cost_threshold=1e-6):
    s, d = _f64(src), _f64(dst)
    R = np.empty(9, np.float64)
    lib().lo_teaser_gnc_rotation(_p(s, C.c_double), _p(d, C.c_double), len(s),
                                 C.c_double(noise_bound), max_iterations, C.c_double(gnc_factor),
                                 C.c_double(cost_threshold), _p(R, C.c_double))
    return R.reshape(3, 3)


def teaser_tls_scalar(x, rng):
    x = _f64(x)
    return lib().lo_teaser_tls_scalar(_p(x, C.c_double), len(x), C.c_double(rng))


def teaser_solve(src, dst, noise_bound=0.5, cbar2=1.0, max_iterations=100, gnc_factor=1.4,
                 cost_threshold=1e-6):
    """RobustRegistrationSolver::solve on matched points [n][3] (double)."""
    s, d = _f64(src), _f64(dst)
    R, t = np.empty(9, np.float64), np.empty(3, np.float64)
    clique = np.empty(max(len(s), 1), np.int32)
    m = C.c_int()
    valid = lib().lo_teaser_solve(_p(s, C.c_double), _p(d, C.c_double), len(s),
                                  C.c_double(noise_bound), C.c_double(cbar2), max_iterations,
                                  C.c_double(gnc_factor), C.c_double(cost_threshold),
                                  _p(R, C.c_double), _p(t, C.c_double), _p(clique, C.c_int32),
                                  C.byref(m))
    return {"valid": bool(valid), "R": R.reshape(3, 3), "t": t, "clique": clique[:m.value].copy()}


def teaser_error_metric(src, dst, R, t, threshold=100.0):
    s, d, R, t = _f64(src), _f64(dst), _f64(R).reshape(9), _f64(t)
    return lib().lo_teaser_error_metric(_p(s, C.c_double), _p(d, C.c_double), len(s),
                                        _p(R, C.c_double), _p(t, C.c_double),
                                        C.c_double(threshold))


def teaser_initial_alignment(src, src_n, tgt, tgt_n, harris_radius=0.1, harris_threshold=5e-9,
                             features_radius=2.0, guess=None):
    """GetTeaserInitialAlignment (IcpLoopComputation.cc:949-1081).  Returns T, used_teaser,
    n_matches, n_clique."""
    src, tgt = _f32(src), _f32(tgt)
    i, _ = harris_keypoints(src, src_n, harris_radius, harris_threshold)
    j, _ = harris_keypoints(tgt, tgt_n, harris_radius, harris_threshold)
    fs = fpfh(src, src_n, src[i], features_radius)
    ft = fpfh(tgt, tgt_n, tgt[j], features_radius)
    T = _f32(np.eye(4) if guess is None else guess).reshape(16).copy()
    sk, tk = _f32(src[i]), _f32(tgt[j])
    nm, nc = C.c_int(), C.c_int()
    used = lib().lo_teaser_initial_alignment(_p(sk, C.c_float), _p(fs, C.c_float), len(sk),
                                             _p(tk, C.c_float), _p(ft, C.c_float), len(tk),
                                             _p(T, C.c_float), C.byref(nm), C.byref(nc))
    return T.reshape(4, 4), bool(used), nm.value, nc.value


def pose_from_quat(q_wxyz, t):
    q = np.ascontiguousarray(q_wxyz, np.float64)
    t = np.ascontiguousarray(t, np.float64)
    p = Pose()
    lib().lo_pose_from_quat(_p(q, C.c_double), _p(t, C.c_double), C.byref(p))
    return p


def pose_between(a, b):
    out = Pose()
    lib().lo_pose_between(C.byref(a), C.byref(b), C.byref(out))
    return out


def pose_inverse(a):
    out = Pose()
    lib().lo_pose_inverse(C.byref(a), C.byref(out))
    return out


def pose_logmap(a):
    xi = np.empty(6, np.float64)
    lib().lo_pose_logmap(C.byref(a), _p(xi, C.c_double))
    return xi


def pose_to_np(p):
    return np.array(p.R, np.float64).reshape(3, 3), np.array(p.t, np.float64)


def base_icp_params(**kw):
    """Reference base-station block (laser_parameters.yaml:205-243,
    lamp/config/precision_parameters.yaml:10-11)."""
    p = IcpParams()
    p.gicp = base_gicp_params()
    p.init_method = INIT_ODOMETRY
    p.covariance_method = COV_POINT2PLANE
    p.max_tolerable_fitness = 0.9
    p.transform_thresholding = 1
    p.max_translation = 20.0
    p.max_rotation_deg = 50.0
    p.rot_sigma = 0.01
    p.trans_sigma = 0.01
    p.use_fixed_covariances = 0
    for k, v in kw.items():
        if hasattr(p.gicp, k) and not hasattr(p, k):
            setattr(p.gicp, k, v)
        else:
            setattr(p, k, v)
    return p


def perform_alignment(params, src_xyz, src_normals, tgt_xyz, tgt_normals, scan1_xyz,
                      scan2_normals, pose_21_odom, pose_21_cand=None):
    a = [_f32(v) for v in (src_xyz, src_normals, tgt_xyz, tgt_normals, scan1_xyz, scan2_normals)]
    out = Alignment()
    if pose_21_cand is None:
        pose_21_cand = pose_21_odom
    lib().lo_perform_alignment(C.byref(params), _p(a[0], C.c_float), _p(a[1], C.c_float), len(a[0]),
                               _p(a[2], C.c_float), _p(a[3], C.c_float), len(a[2]),
                               _p(a[4], C.c_float), len(a[4]), _p(a[5], C.c_float), len(a[5]),
                               C.byref(pose_21_odom), C.byref(pose_21_cand), C.byref(out))
    R, t = pose_to_np(out.delta)
    return {
        "status": out.status,
        "T_icp": np.array(out.T_icp, np.float32).reshape(4, 4),
        "delta_R": R,
        "delta_t": t,
        "fitness": out.fitness,
        "covariance": np.array(out.covariance, np.float64).reshape(6, 6),
        "iterations": out.iterations,
        "converged": bool(out.converged),
    }


def mat3_inverse(a):
    a = np.ascontiguousarray(a, np.float64).reshape(9)
    out = np.empty(9, np.float64)
    lib().lo_mat3_inverse(_p(a, C.c_double), _p(out, C.c_double))
    return out.reshape(3, 3)


def mat6_inverse(a):
    a = np.ascontiguousarray(a, np.float64).reshape(36)
    out = np.empty(36, np.float64)
    lib().lo_mat6_inverse(_p(a, C.c_double), _p(out, C.c_double))
    return out.reshape(6, 6)


def sym_eig(a):
    a = np.ascontiguousarray(a, np.float64)
    n = a.shape[0]
    ev = np.empty(n, np.float64)
    V = np.empty(n * n, np.float64)
    lib().lo_sym_eig(_p(a.reshape(-1), C.c_double), n, _p(ev, C.c_double), _p(V, C.c_double))
    return ev, V.reshape(n, n)


def ldlt(a):
    a = np.ascontiguousarray(a, np.float64).reshape(36)
    L = np.empty(36, np.float64)
    D = np.empty(6, np.float64)
    tr = np.empty(6, np.int32)
    lib().lo_ldlt(_p(a, C.c_double), _p(L, C.c_double), _p(D, C.c_double), _p(tr, C.c_int))
    return L.reshape(6, 6), D, tr
