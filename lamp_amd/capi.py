"""ctypes binding of include/lamp_gicp.h (the C ABI of liblamp_gicp.so)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblamp_gicp.so")

ABI_VERSION = 3
INIT_IDENTITY, INIT_ODOMETRY, INIT_ODOM_ROTATION, INIT_FEATURES, INIT_TEASERPP, INIT_CANDIDATE = range(6)
COV_POINT2POINT, COV_POINT2PLANE = 0, 1
SOLVER_GAUSS_NEWTON, SOLVER_BFGS = 0, 1
(PAIR_OK, PAIR_MISSING_SCAN, PAIR_MISSING_POSE, PAIR_EMPTY_CLOUD, PAIR_INIT_REJECTED,
 PAIR_NONFINITE, PAIR_NOT_CONVERGED, PAIR_FITNESS, PAIR_TRANSFORM) = range(9)

EXPORTS = ["lgicp_default_config", "lgicp_create", "lgicp_destroy", "lgicp_last_error",
           "lgicp_add_keyframe", "lgicp_add_pose", "lgicp_has_keyframe", "lgicp_has_pose",
           "lgicp_submit", "lgicp_submit_stats", "lgicp_get_stats", "lgicp_nn_search", "lgicp_covariances_from_normals",
           "lgicp_covariances_knn", "lgicp_align", "lgicp_compute_ap", "lgicp_fpfh",
           "lgicp_harris_keypoints", "lgicp_sac_initial_alignment",
           "lgicp_teaser_initial_alignment", "lgicp_compute_normals", "lgicp_icp_observability",
           "lgicp_voxel_grid", "lgicp_random_sample", "lgicp_nn_search_static", "lgicp_add_keyframes", "lgicp_wait_registered",
           "lgicp_set_submit_lanes", "lgicp_teaser_clique"]


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int), ("device", C.c_int), ("tf_epsilon", C.c_double),
                ("corr_dist", C.c_double), ("iterations", C.c_int),
                ("optimizer_iterations", C.c_int), ("rotation_epsilon", C.c_double),
                ("k_correspondences", C.c_int), ("gicp_epsilon", C.c_double),
                ("covariance_from_knn", C.c_int), ("init_method", C.c_int),
                ("covariance_method", C.c_int), ("max_tolerable_fitness", C.c_double),
                ("transform_thresholding", C.c_int), ("max_translation", C.c_double),
                ("max_rotation_deg", C.c_double), ("num_prev_scans", C.c_int),
                ("num_next_scans", C.c_int), ("accumulate_source", C.c_int),
                ("laser_lc_rot_sigma", C.c_double), ("laser_lc_trans_sigma", C.c_double),
                ("use_fixed_covariances", C.c_int), ("sac_iterations", C.c_int),
                ("sac_features_radius", C.c_double), ("sac_fitness_score_threshold", C.c_double),
                ("harris_threshold", C.c_double), ("harris_suppression", C.c_int),
                ("harris_radius", C.c_double), ("harris_refine", C.c_int),
                ("harris_response", C.c_int), ("workspace_bytes", C.c_size_t),
                ("accumulated_cache_bytes", C.c_size_t), ("solver", C.c_int), ("stage_timing", C.c_int), ("reserved", C.c_int * 8)]


class Candidate(C.Structure):
    _fields_ = [("key_from", C.c_uint64), ("key_to", C.c_uint64), ("pose_from_q", C.c_double * 4),
                ("pose_from_t", C.c_double * 3), ("pose_to_q", C.c_double * 4),
                ("pose_to_t", C.c_double * 3)]


class Result(C.Structure):
    _fields_ = [("key_from", C.c_uint64), ("key_to", C.c_uint64), ("status", C.c_int32),
                ("iterations", C.c_int32), ("converged", C.c_int32),
                ("n_correspondences", C.c_int32), ("T_icp", C.c_float * 16),
                ("delta_R", C.c_double * 9), ("delta_t", C.c_double * 3),
                ("delta_q", C.c_double * 4), ("fitness", C.c_double),
                ("covariance", C.c_double * 36), ("information", C.c_double * 36),
                ("T_init", C.c_float * 16), ("init_fitness", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("pairs_submitted", C.c_uint64), ("pairs_ok", C.c_uint64),
                ("outer_iterations", C.c_uint64), ("nn_launches", C.c_uint64),
                ("nn_distance_evals", C.c_double), ("nn_ms", C.c_double), ("reduce_ms", C.c_double),
                ("solve_ms", C.c_double), ("cov_ms", C.c_double), ("post_ms", C.c_double),
                ("total_ms", C.c_double), ("last_nn_kernel_ms", C.c_double),
                ("last_nn_evals_per_launch", C.c_double), ("nn_executed_evals", C.c_double),
                ("reduce_points", C.c_double), ("reduce_inliers", C.c_double),
                ("reduce_launches", C.c_uint64), ("cycle_exits", C.c_uint64),
                ("clique_ms", C.c_double), ("clique_row_bytes", C.c_double),
                ("clique_launches", C.c_uint64), ("core_launches", C.c_uint64),
                ("core_ms", C.c_double), ("core_matrix_bytes", C.c_double),
                ("teaser_oversized_pairs", C.c_uint64)]


class LgicpError(RuntimeError):
    pass


_lib = None


def lib():
    """Load liblamp_gicp.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LgicpError(
                f"{LIB_PATH} is missing: build it with `make -C lamp_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.lgicp_last_error.restype = C.c_char_p
        L.lgicp_last_error.argtypes = [C.c_void_p]
        L.lgicp_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
        L.lgicp_destroy.argtypes = [C.c_void_p]
        L.lgicp_add_keyframe.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                         C.c_size_t, C.c_size_t]
        L.lgicp_add_keyframes.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_size_t]
        L.lgicp_wait_registered.argtypes = [C.c_void_p]
        L.lgicp_add_pose.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.lgicp_has_keyframe.argtypes = [C.c_void_p, C.c_uint64]
        L.lgicp_has_pose.argtypes = [C.c_void_p, C.c_uint64]
        L.lgicp_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.lgicp_submit_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.lgicp_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.lgicp_set_submit_lanes.argtypes = [C.c_void_p, C.c_int]
        L.lgicp_nn_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                      C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
        L.lgicp_covariances_from_normals.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.lgicp_covariances_knn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                            C.c_double, C.c_void_p]
        L.lgicp_align.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                  C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                  C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.lgicp_compute_ap.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                       C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]
        L.lgicp_fpfh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                 C.c_size_t, C.c_double, C.c_void_p]
        L.lgicp_harris_keypoints.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                             C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.lgicp_sac_initial_alignment.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                                  C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                                  C.c_void_p, C.POINTER(C.c_double)]
        L.lgicp_teaser_initial_alignment.argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
            C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
            C.POINTER(C.c_int32)]
        L.lgicp_teaser_clique.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                          C.POINTER(C.c_int32)]
        L.lgicp_compute_normals.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p,
                                            C.c_void_p]
        L.lgicp_icp_observability.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                              C.c_void_p]
        L.lgicp_voxel_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_double,
                                       C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
        L.lgicp_random_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                          C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
        L.lgicp_nn_search_static.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                             C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                             C.c_void_p]
        _lib = L
    return _lib


def default_config(base_station=True, **kw):
    cfg = Config()
    lib().lgicp_default_config(C.byref(cfg), 1 if base_station else 0)
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """Owns one lgicp_ctx (one GPU, one stream)."""

    def __init__(self, cfg=None, **kw):
        self.cfg = cfg if cfg is not None else default_config(**kw)
        self._h = C.c_void_p()
        rc = lib().lgicp_create(C.byref(self.cfg), C.byref(self._h))
        if rc != 0:
            raise LgicpError(f"lgicp_create failed ({rc}): {lib().lgicp_last_error(None).decode()}")

    def close(self):
        if self._h:
            lib().lgicp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise LgicpError(f"lgicp error {rc}: {lib().lgicp_last_error(self._h).decode()}")
        return rc

    def add_keyframe(self, key, xyz, normals=None):
        xyz = _f32(xyz).reshape(-1, 3)
        nrm = None if normals is None else _f32(normals).reshape(-1, 3)
        return self._check(lib().lgicp_add_keyframe(self._h, key, _ptr(xyz),
                                                    None if nrm is None else _ptr(nrm), len(xyz), 12))

    def add_keyframes(self, keys, clouds, normals=None):
        """lgicp_add_keyframes: clouds / normals are lists of [n, 3] arrays (normals or single
        entries of it may be None).  Returns the number of scans stored."""
        n = len(keys)
        xs = [_f32(c).reshape(-1, 3) for c in clouds]
        ns = [None if (normals is None or normals[k] is None) else _f32(normals[k]).reshape(-1, 3)
              for k in range(n)]
        karr = (C.c_uint64 * max(n, 1))(*keys)
        xarr = (C.c_void_p * max(n, 1))(*[x.ctypes.data for x in xs])
        narr = (C.c_void_p * max(n, 1))(*[None if v is None else v.ctypes.data for v in ns])
        cnt = (C.c_size_t * max(n, 1))(*[len(x) for x in xs])
        return self._check(lib().lgicp_add_keyframes(self._h, n, karr, xarr, narr, cnt, 12))

    def wait_registered(self):
        return self._check(lib().lgicp_wait_registered(self._h))

    def add_pose(self, key, q_wxyz, t):
        q = np.ascontiguousarray(q_wxyz, np.float64)
        t = np.ascontiguousarray(t, np.float64)
        return self._check(lib().lgicp_add_pose(self._h, key, _ptr(q), _ptr(t)))

    def has_keyframe(self, key):
        return bool(lib().lgicp_has_keyframe(self._h, key))

    def has_pose(self, key):
        return bool(lib().lgicp_has_pose(self._h, key))

    def pack_candidates(self, candidates):
        """The lgicp_candidate array a C++ caller would hand over as is."""
        n = len(candidates)
        arr = (Candidate * max(n, 1))()
        for i, c in enumerate(candidates):
            arr[i].key_from, arr[i].key_to = c[0], c[1]
            pf = c[2] if len(c) > 2 else ((1, 0, 0, 0), (0, 0, 0))
            pt = c[3] if len(c) > 3 else ((1, 0, 0, 0), (0, 0, 0))
            arr[i].pose_from_q[:] = pf[0]
            arr[i].pose_from_t[:] = pf[1]
            arr[i].pose_to_q[:] = pt[0]
            arr[i].pose_to_t[:] = pt[1]
        return arr, n, (Result * max(n, 1))()

    def submit_packed(self, packed):
        """lgicp_submit on arrays from pack_candidates; returns the lgicp_result array."""
        arr, n, out = packed
        self._check(lib().lgicp_submit(self._h, arr, n, out))
        return out

    def submit_packed_stats(self, packed):
        """lgicp_submit_stats: the lgicp_result array and the statistics of this very submit."""
        arr, n, out = packed
        st = Stats()
        self._check(lib().lgicp_submit_stats(self._h, arr, n, out, C.byref(st)))
        return out, st

    def set_submit_lanes(self, lanes):
        """lgicp_set_submit_lanes: submits of different threads that may run side by side (1..8; at
        most 2 for a context with a feature-based initialisation); returns the number in effect."""
        return self._check(lib().lgicp_set_submit_lanes(self._h, lanes))

    def submit(self, candidates):
        """candidates: list of (key_from, key_to[, pose_from(q,t), pose_to(q,t)])"""
        n = len(candidates)
        arr = (Candidate * max(n, 1))()
        for i, c in enumerate(candidates):
            arr[i].key_from, arr[i].key_to = c[0], c[1]
            pf = c[2] if len(c) > 2 else ((1, 0, 0, 0), (0, 0, 0))
            pt = c[3] if len(c) > 3 else ((1, 0, 0, 0), (0, 0, 0))
            arr[i].pose_from_q[:] = pf[0]
            arr[i].pose_from_t[:] = pf[1]
            arr[i].pose_to_q[:] = pt[0]
            arr[i].pose_to_t[:] = pt[1]
        out = (Result * max(n, 1))()
        self._check(lib().lgicp_submit(self._h, arr, n, out))
        return [out[i] for i in range(n)]

    def stats(self):
        s = Stats()
        self._check(lib().lgicp_get_stats(self._h, C.byref(s)))
        return s

    def nn_search(self, T, src, tgt, n_splits=0):
        T = _f32(T).reshape(16)
        src, tgt = _f32(src).reshape(-1, 3), _f32(tgt).reshape(-1, 3)
        idx = np.empty(len(src), np.int32)
        d2 = np.empty(len(src), np.float32)
        self._check(lib().lgicp_nn_search(self._h, _ptr(T), _ptr(src), len(src), _ptr(tgt),
                                          len(tgt), _ptr(idx), _ptr(d2), n_splits))
        return idx, d2

    def nn_search_static(self, T, d_src, n_src, d_tgt, n_tgt, d_idx, d_d2, stream=None):
        """lgicp_nn_search_static: the d_* arguments are device addresses (ints)."""
        T = _f32(T).reshape(16)
        return self._check(lib().lgicp_nn_search_static(
            self._h, _ptr(T), C.c_void_p(d_src), n_src, C.c_void_p(d_tgt), n_tgt,
            C.c_void_p(d_idx), C.c_void_p(d_d2), C.c_void_p(stream) if stream else None))

    def covariances_from_normals(self, normals):
        nrm = _f32(normals).reshape(-1, 3)
        out = np.empty((len(nrm), 3, 3), np.float64)
        self._check(lib().lgicp_covariances_from_normals(self._h, _ptr(nrm), len(nrm), _ptr(out)))
        return out

    def covariances_knn(self, xyz, k=20, eps=1e-3):
        xyz = _f32(xyz).reshape(-1, 3)
        out = np.empty((len(xyz), 3, 3), np.float64)
        self._check(lib().lgicp_covariances_knn(self._h, _ptr(xyz), len(xyz), k, eps, _ptr(out)))
        return out

    def align(self, src, src_n, tgt, tgt_n, guess=None):
        src, src_n = _f32(src).reshape(-1, 3), _f32(src_n).reshape(-1, 3)
        tgt, tgt_n = _f32(tgt).reshape(-1, 3), _f32(tgt_n).reshape(-1, 3)
        g = _f32(np.eye(4) if guess is None else guess).reshape(16)
        T = np.empty(16, np.float32)
        conv, it, fit = C.c_int(), C.c_int(), C.c_double()
        self._check(lib().lgicp_align(self._h, _ptr(src), _ptr(src_n), len(src), _ptr(tgt),
                                      _ptr(tgt_n), len(tgt), _ptr(g), _ptr(T), C.byref(conv),
                                      C.byref(it), C.byref(fit)))
        return {"T": T.reshape(4, 4), "converged": bool(conv.value), "iterations": it.value,
                "fitness": fit.value}

    def fpfh(self, surface_xyz, surface_normals, keypoints_xyz, radius):
        sx, sn = _f32(surface_xyz).reshape(-1, 3), _f32(surface_normals).reshape(-1, 3)
        kx = _f32(keypoints_xyz).reshape(-1, 3)
        out = np.empty((len(kx), 33), np.float32)
        self._check(lib().lgicp_fpfh(self._h, _ptr(sx), _ptr(sn), len(sx), _ptr(kx), len(kx),
                                     float(radius), _ptr(out)))
        return out

    def harris_keypoints(self, xyz, normals=None):
        """normals=None: the ComputeKeypoints overload that estimates them over the Harris radius"""
        x = _f32(xyz).reshape(-1, 3)
        nn = None if normals is None else _f32(normals).reshape(-1, 3)
        idx = np.empty(len(x), np.int32)
        cnt = C.c_size_t()
        self._check(lib().lgicp_harris_keypoints(self._h, _ptr(x), None if nn is None else _ptr(nn),
                                                 len(x), _ptr(idx), len(idx), C.byref(cnt)))
        return idx[: cnt.value].copy()

    def sac_initial_alignment(self, src, src_n, tgt, tgt_n, guess=None):
        a = [_f32(v).reshape(-1, 3) for v in (src, src_n, tgt, tgt_n)]
        g = _f32(np.eye(4) if guess is None else guess).reshape(16)
        T = np.empty(16, np.float32)
        fit = C.c_double()
        self._check(lib().lgicp_sac_initial_alignment(self._h, _ptr(a[0]), _ptr(a[1]), len(a[0]),
                                                      _ptr(a[2]), _ptr(a[3]), len(a[2]), _ptr(g),
                                                      _ptr(T), C.byref(fit)))
        return T.reshape(4, 4), fit.value

    def teaser_initial_alignment(self, src, src_n, tgt, tgt_n, guess=None):
        a = [_f32(v).reshape(-1, 3) for v in (src, src_n, tgt, tgt_n)]
        g = _f32(np.eye(4) if guess is None else guess).reshape(16)
        T = np.empty(16, np.float32)
        used, nm, nc = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(lib().lgicp_teaser_initial_alignment(
            self._h, _ptr(a[0]), _ptr(a[1]), len(a[0]), _ptr(a[2]), _ptr(a[3]), len(a[2]), _ptr(g),
            _ptr(T), C.byref(used), C.byref(nm), C.byref(nc)))
        return T.reshape(4, 4), bool(used.value), nm.value, nc.value

    def teaser_clique(self, adj):
        """adj: n x n boolean adjacency (symmetric, no self loops).  The clique as sorted indices."""
        adj = np.asarray(adj, bool)
        n = len(adj)
        words = (n + 63) // 64
        padded = np.zeros((n, words * 64), np.uint8)
        padded[:, :n] = adj
        bits = np.ascontiguousarray(np.packbits(padded, axis=1, bitorder="little")).view("<u8")
        out = np.empty(max(n, 1), np.int32)
        m = C.c_int32()
        self._check(lib().lgicp_teaser_clique(self._h, _ptr(bits), n, _ptr(out), C.byref(m)))
        return out[:m.value].copy()

    def compute_normals(self, xyz, k=10):
        x = _f32(xyz).reshape(-1, 3)
        nrm = np.empty((len(x), 3), np.float32)
        cur = np.empty(len(x), np.float32)
        self._check(lib().lgicp_compute_normals(self._h, _ptr(x), len(x), int(k), _ptr(nrm), _ptr(cur)))
        return nrm, cur

    def icp_observability(self, xyz, normals):
        x, nn = _f32(xyz).reshape(-1, 3), _f32(normals).reshape(-1, 3)
        ev = np.empty(3, np.float64)
        self._check(lib().lgicp_icp_observability(self._h, _ptr(x), _ptr(nn), len(x), _ptr(ev)))
        return ev

    def voxel_grid(self, xyz, leaf_size, normals=None):
        """pcl::VoxelGrid with a cubic leaf.  Returns xyz, or (xyz, normals) when normals are given."""
        x = _f32(xyz).reshape(-1, 3)
        nn = None if normals is None else _f32(normals).reshape(-1, 3)
        out = np.empty((len(x), 3), np.float32)
        out_n = None if nn is None else np.empty((len(x), 3), np.float32)
        m = C.c_size_t(0)
        self._check(lib().lgicp_voxel_grid(self._h, _ptr(x), None if nn is None else _ptr(nn), len(x),
                                           C.c_double(leaf_size), _ptr(out),
                                           None if out_n is None else _ptr(out_n), C.byref(m)))
        return out[:m.value].copy() if nn is None else (out[:m.value].copy(), out_n[:m.value].copy())

    def random_sample(self, xyz, n_keep, seed=0, normals=None):
        """pcl::RandomSample: min(n_keep, n) points in input order, drawn from `seed`."""
        x = _f32(xyz).reshape(-1, 3)
        nn = None if normals is None else _f32(normals).reshape(-1, 3)
        out = np.empty((len(x), 3), np.float32)
        out_n = None if nn is None else np.empty((len(x), 3), np.float32)
        m = C.c_size_t(0)
        self._check(lib().lgicp_random_sample(self._h, _ptr(x), None if nn is None else _ptr(nn), len(x),
                                              C.c_size_t(max(int(n_keep), 0)), C.c_uint32(seed & 0xFFFFFFFF),
                                              _ptr(out), None if out_n is None else _ptr(out_n),
                                              C.byref(m)))
        return out[:m.value].copy() if nn is None else (out[:m.value].copy(), out_n[:m.value].copy())

    def compute_ap(self, query, ref_normals, corr, T, normalize=True):
        q, rn = _f32(query).reshape(-1, 3), _f32(ref_normals).reshape(-1, 3)
        corr = np.ascontiguousarray(corr, np.int32)
        T = _f32(T).reshape(16)
        Ap = np.empty(36, np.float64)
        self._check(lib().lgicp_compute_ap(self._h, _ptr(q), len(q), _ptr(rn), len(rn), _ptr(corr),
                                           len(corr), _ptr(T), int(normalize), _ptr(Ap)))
        return Ap.reshape(6, 6)


def pair_seed(key_from, key_to):
    """Sampling seed lgicp_submit gives the SAC-IA of one candidate."""
    return (0x5AC1A000 ^ ((key_from * 2654435761) & 0xFFFFFFFF) ^ ((key_to * 40503) & 0xFFFFFFFF)) \
        & 0xFFFFFFFF


def result_to_dict(r):
    return {
        "key_from": r.key_from, "key_to": r.key_to, "status": r.status,
        "iterations": r.iterations, "converged": bool(r.converged),
        "n_correspondences": r.n_correspondences,
        "T_icp": np.array(r.T_icp, np.float32).reshape(4, 4),
        "delta_R": np.array(r.delta_R).reshape(3, 3), "delta_t": np.array(r.delta_t),
        "delta_q": np.array(r.delta_q), "fitness": r.fitness,
        "covariance": np.array(r.covariance).reshape(6, 6),
        "information": np.array(r.information).reshape(6, 6),
        "T_init": np.array(r.T_init, dtype=np.float32).reshape(4, 4),
        "init_fitness": r.init_fitness,
    }
