/*
 * lamp_oracle.h — CPU restatement of LAMP's loop-closure geometric verification path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / the timed CPU baseline.  The HIP library (lamp_amd/csrc) never
 * links or calls it.
 *
 * What it restates (reference file:line given at every function):
 *   lamp_utils/include/lamp_utils/gicp.hpp, gicp.h, gicp_utils.h   (multithreaded GICP)
 *   lamp_utils/src/PointCloudUtils.cc                              (normalise, Ap, observability)
 *   loop_closure/src/IcpLoopComputation.cc                         (PerformAlignment & friends)
 *
 * Third-party arithmetic that is NOT under /root/reference and is restated from its
 * published algorithm: PCL >= 1.8 (Registration::align, getFitnessScore,
 * transformPointCloud, pcl::BFGS = port of GSL vector_bfgs2 / Fletcher line search),
 * FLANN KDTreeSingleIndex<L2_Simple> (exact 1-NN, float squared distance accumulated
 * x,y,z in order), Eigen (3x3 cofactor inverse, JacobiSVD, LDLT with diagonal pivoting),
 * GTSAM Pose3 (between, Logmap), geometry_utils (PoseInverse).
 *
 * Parity pins (reference tests): Ap(0,0)=Ap(1,1)=56.77534, Ap(5,5)=100; observability
 * eigenvalues (0,0,100); NormalizePCloud moments; plane normals; corner-vs-shifted-corner
 * alignment recovers T=(1,-0.001,0) to 1e-3.  NOT pinned by any reference fixture (parity
 * unpinned): the BFGS iterate sequence, fitness values, covariance values of a real
 * alignment, FPFH values.
 *
 * All matrices are ROW-MAJOR in this API.  Points are packed float xyz (stride 3).
 */
#ifndef LAMP_ORACLE_H_
#define LAMP_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ transforms */
/* pcl::transformPointCloud with a float matrix: ((m00*x+m01*y)+m02*z)+m03, no FMA. */
void lo_transform_points_f32(const float T[16], const float* xyz, int n, float* out);
/* same with a double matrix, result cast to float (AccumulateScans passes a gtsam Matrix4). */
void lo_transform_points_f64(const double T[16], const float* xyz, int n, float* out);
/* Eigen fixed-size 4x4 float product a*b (k = 0..3 in order, no FMA). */
void lo_mat4f_mul(const float a[16], const float b[16], float out[16]);

/* ------------------------------------------------------------------ nearest neighbours */
/* Exhaustive 1-NN, d = (dx*dx + dy*dy) + dz*dz in float; lowest target index wins ties.
 * idx[i] = -1 and d2[i] = +inf when nt == 0 or every distance is NaN. */
void lo_nn_brute(const float* q, int nq, const float* t, int nt, int32_t* idx, float* d2);
/* Exhaustive k-NN (ascending distance, lowest index first on ties). */
void lo_knn_brute(const float* q, int nq, const float* t, int nt, int k, int32_t* idx, float* d2);

typedef struct lo_kdtree lo_kdtree;
lo_kdtree* lo_kdtree_build(const float* xyz, int n);
void lo_kdtree_free(lo_kdtree* t);
/* exact; same distance arithmetic and the same lowest-index tie rule as lo_nn_brute */
void lo_kdtree_nn(const lo_kdtree* t, const float* q, int nq, int32_t* idx, float* d2, int threads);
void lo_kdtree_knn(const lo_kdtree* t, const float* q, int nq, int k, int32_t* idx, float* d2,
                   int threads);

/* ------------------------------------------------------------------ covariances */
/* gicp_utils.h:28-62 */
void lo_cov_from_normals(const float* normals, int n, double* cov9);
/* gicp.hpp:83-155 (k-NN in own cloud, divide by k, SVD -> (1,1,eps)). returns 0 on success,
 * -1 when k > n (gicp.hpp:72-79: covariances are then left empty). */
int lo_cov_knn(const float* xyz, int n, int k, double eps, double* cov9, int threads);

/* ------------------------------------------------------------------ GICP */
enum { LO_SOLVER_BFGS = 0, LO_SOLVER_GN = 1 };

typedef struct lo_gicp_params {
  double corr_dist;          /* setMaxCorrespondenceDistance  (IcpLoopComputation.cc:196) */
  double tf_epsilon;         /* setTransformationEpsilon      (IcpLoopComputation.cc:195) */
  double rotation_epsilon;   /* gicp.h:128  (2e-3)                                          */
  int max_iterations;        /* setMaximumIterations          (IcpLoopComputation.cc:197) */
  int max_inner_iterations;  /* setMaximumOptimizerIterations (IcpLoopComputation.cc:199) */
  int k_correspondences;     /* gicp.h:121 (20)                                             */
  double gicp_epsilon;       /* gicp.h:127 (1e-3)                                           */
  int cov_from_knn;          /* 0: covariances from normals (default path, gicp.hpp:81-82) */
  int solver;                /* LO_SOLVER_BFGS (reference) | LO_SOLVER_GN (what the GPU runs) */
  int use_kdtree;            /* 1: kd-tree lookups (reference-like cost), 0: brute force    */
  int threads;               /* omp threads of the lookup loop (gicp.hpp:463-465)           */
  /* measurement knobs, not in the reference */
  double gradient_tol;       /* BFGS gradient tolerance; the reference hard-codes 1e-2 (gicp.hpp:250) */
  int gn_cycle_exit;         /* 1 (default): the Gauss-Newton back end's limit-cycle exit; 0: off */
  int residual_double;       /* 0 (default, the reference): BFGS functor rounds T*p to float
                                (gicp.hpp:307-310); 1: residual in double from the float matrix */
  int sum_order;             /* 0 (default, the reference): the functor adds the correspondences one
                                after the other (gicp.hpp:298-315); 1: the order of a 512-lane
                                parallel sum (lane = source index mod 512, a butterfly over each
                                group of 64 lanes, the 8 groups in turn) -- what a different but
                                equally valid order of the same additions does to the answer */
} lo_gicp_params;

void lo_gicp_default_params(lo_gicp_params* p);

typedef struct lo_gicp_result {
  float final_transformation[16]; /* previous_ * guess (gicp.hpp:583)                    */
  int converged;                  /* converged_                                           */
  int iterations;                 /* nr_iterations_                                       */
  int last_num_correspondences;   /* m of the last outer iteration                        */
  int failure;                    /* 1: a covariance set was empty (cloud smaller than k) */
  double last_delta;
  int cycle_exit;                 /* 1: left the loop through the Gauss-Newton cycle exit */
  int pad_;
} lo_gicp_result;

/* MultithreadedGICP::computeTransformation (gicp.hpp:405-617) driven as Registration::align.
 * trace_idx (nullable): int32 [trace_cap][n_src] target index per outer iteration (-1 = gated)
 * trace_T   (nullable): float [trace_cap][16] transformation_ used for the lookups          */
int lo_gicp_align(const lo_gicp_params* p, const float* src_xyz, const float* src_normals,
                  int n_src, const float* tgt_xyz, const float* tgt_normals, int n_tgt,
                  const float guess[16], lo_gicp_result* out, int32_t* trace_idx, float* trace_T,
                  int trace_cap);

/* measurement helper (not a reference function): objective and gradient of gicp.hpp:290-402 at the
 * stationary point T claims to be — lookups and Mahalanobis matrices of one outer iteration started
 * from T, functor evaluated at x = 0, residual in double.  Returns -1 with fewer than 4 matches. */
int lo_gicp_objective_at(const lo_gicp_params* p, const float* src_xyz, const float* src_normals,
                         int n_src, const float* tgt_xyz, const float* tgt_normals, int n_tgt,
                         const float T[16], double* f, double g[6], int* m_out);

/* one evaluation of the BFGS functor (gicp.hpp:290-402) for unit tests:
 * f and g[6] at x for given correspondences; mahal9 is [n_src][9] indexed by source index. */
void lo_gicp_cost_grad(const float* src_xyz_guess, const float* tgt_xyz, const int32_t* src_idx,
                       const int32_t* tgt_idx, int m, const double* mahal9, const double x[6],
                       double* f, double g[6]);
/* gicp.hpp:619-634 applied to identity */
void lo_apply_state(const double x[6], float T[16]);

/* Registration::getFitnessScore (PCL): mean squared 1-NN distance of final*src in tgt.
 * idx_out (nullable) receives the 1-NN index of every transformed point
 * (IcpLoopComputation.cc:553-566). */
double lo_fitness_score(const float* src_xyz, int n_src, const float* tgt_xyz, int n_tgt,
                        const float final_T[16], int32_t* idx_out, int use_kdtree, int threads);

/* ------------------------------------------------------------------ PointCloudUtils */
/* PointCloudUtils.cc:39-59 */
void lo_normalize_cloud(const float* xyz, int n, float* out);
/* PointCloudUtils.cc:115-163; corr[i] >= n_ref_normals -> zero normal */
void lo_compute_ap(const float* query_normalized, int n_query, const float* ref_normals,
                   int n_ref_normals, const int64_t* corr, int n_corr, const float T[16],
                   double Ap[36]);
/* PointCloudUtils.cc:165-196 (normals taken from the cloud; ascending eigenvalues) */
void lo_icp_observability(const float* xyz, const float* normals, int n, double eig[3]);

/* ------------------------------------------------------------------ FPFH */
/* lamp_utils::ComputeFeatures (PointCloudUtils.cc:98-113): PCL FPFHEstimationOMP, input =
 * keypoints, surface = full cloud.  out is [n_key][33] floats = [f1 | f2 | f3] */
void lo_fpfh(const float* surface_xyz, const float* surface_normals, int n,
             const float* keypoints_xyz, int n_key, double radius, float* out);

/* ------------------------------------------------------------------ Harris3D + SAC-IA */
/* pcl::HarrisKeypoint3D (HARRIS response) as configured by lamp_utils::ComputeKeypoints
 * (PointCloudUtils.cc:78-96) */
void lo_harris_response(const float* xyz, const float* normals, int n, double radius,
                        float* response);
int lo_harris_keypoints(const float* xyz, const float* normals, int n, double radius,
                        double threshold, int nms, int32_t* out_idx, float* out_response);
/* pcl::SampleConsensusInitialAlignment as driven by GetSacInitialAlignment
 * (IcpLoopComputation.cc:650-705); returns getFitnessScore() */
double lo_sac_ia(const float* src_kp, const float* src_feat, int n_src, const float* tgt_kp,
                 const float* tgt_feat, int n_tgt, int max_iterations, int k_corr,
                 const float guess[16], uint32_t seed, float T_out[16]);

/* ------------------------------------------------------------------ normal estimation */
/* lamp_utils::ComputeNormals (PointCloudUtils.h:38-62), k-NN search; normals [n][3] */
void lo_point_normal(const float* xyz, const int32_t* idx, int k, const float* query, float nrm[3],
                     float* curvature);
void lo_compute_normals(const float* xyz, int n, int k, float* normals, float* curvature,
                        int threads);
/* pcl::NormalEstimation over a radius search: what pcl::HarrisKeypoint3D computes for itself in
 * the lamp_utils::ComputeKeypoints overload without normals (PointCloudUtils.cc:61-76) */
void lo_compute_normals_radius(const float* xyz, int n, double radius, float* normals,
                               float* curvature);
/* lamp_utils::ExtractNormals (PointCloudUtils.cc:17-35) */
void lo_extract_normals(const float* xyz, const float* stored, int n, int k, float* normals,
                        int threads);

/* ------------------------------------------------------------------ down-sampling filters */
/* pcl::VoxelGrid, cubic leaf, all fields (LampPcldFilter.cc:46-50); returns the output count.
 * normals / out_normals may be NULL.  out_* hold up to n points. */
size_t lo_voxel_grid(const float* xyz, const float* normals, size_t n, double leaf_size,
                     float* out_xyz, float* out_normals);
/* pcl::RandomSample's contract (LampPcldFilter.cc:26-33) with a seeded draw; ascending indices */
size_t lo_random_sample(size_t n, size_t n_keep, uint32_t seed, int32_t* out_indices);

/* ------------------------------------------------------------------ TEASER++ (parity unpinned) */
/* GetTeaserInitialAlignment (IcpLoopComputation.cc:949-1081) from the keypoints on; see
 * lo_teaser.c for the restated algorithm.  pairs is [count][2] = (source, target) keypoint ids */
int lo_teaser_matches(const float* src_feat, int n_src, const float* tgt_feat, int n_tgt, int k,
                      int32_t* pairs);
int lo_teaser_clique(const uint8_t* adj, int n, int32_t* clique);
void lo_teaser_gnc_rotation(const double* src, const double* dst, int n, double noise_bound,
                            int max_iterations, double gnc_factor, double cost_threshold,
                            double R[9]);
double lo_teaser_tls_scalar(const double* X, int n, double range);
int lo_teaser_solve(const double* src, const double* dst, int n, double noise_bound, double cbar2,
                    int rotation_max_iterations, double gnc_factor, double cost_threshold,
                    double R[9], double t[3], int32_t* clique_out, int* n_clique_out);
double lo_teaser_error_metric(const double* src, const double* dst, int n, const double R[9],
                              const double t[3], double threshold);
int lo_teaser_initial_alignment(const float* src_kp, const float* src_feat, int n_src,
                                const float* tgt_kp, const float* tgt_feat, int n_tgt, float T[16],
                                int* n_matches_out, int* n_clique_out);

/* ------------------------------------------------------------------ IcpLoopComputation */
/* IcpLoopComputation.cc:707-768 */
void lo_icp_cov_point_plane(const float* scan1_xyz, int n1, const float* scan2_normals, int n2,
                            const int64_t* corr, int n_corr, const float T[16], double rot_sigma,
                            double trans_sigma, double cov[36], double Ap_out[36]);
/* IcpLoopComputation.cc:770-908 */
int lo_icp_cov_point_point(const float* aligned_xyz, int n, const float T[16], double fitness,
                           double cov[36]);

typedef struct lo_pose {
  double R[9];
  double t[3];
} lo_pose;
/* gtsam::Rot3(w,x,y,z) = Eigen quaternion -> matrix, NOT normalised (a zero quaternion gives I) */
void lo_pose_from_quat(const double q_wxyz[4], const double t[3], lo_pose* out);
void lo_pose_between(const lo_pose* a, const lo_pose* b, lo_pose* out); /* a^-1 * b */
void lo_pose_inverse(const lo_pose* a, lo_pose* out);                   /* gu::PoseInverse */
void lo_pose_logmap(const lo_pose* a, double xi[6]);                    /* gtsam Pose3::Logmap (rot,trans) */

enum {
  LO_INIT_IDENTITY = 0,
  LO_INIT_ODOMETRY = 1,
  LO_INIT_ODOM_ROTATION = 2,
  LO_INIT_FEATURES = 3,
  LO_INIT_TEASERPP = 4,
  LO_INIT_CANDIDATE = 5
};
enum { LO_COV_POINT2POINT = 0, LO_COV_POINT2PLANE = 1 };

enum {
  LO_OK = 0,
  LO_FAIL_EMPTY = 1,          /* IcpLoopComputation.cc:458-465 */
  LO_FAIL_NONFINITE = 2,      /* :558-559 */
  LO_FAIL_NOT_CONVERGED = 3,  /* :580-584 */
  LO_FAIL_FITNESS = 4,        /* :588-593 */
  LO_FAIL_TRANSFORM = 5,      /* :604-614 */
  LO_FAIL_UNSUPPORTED_INIT = 6
};

typedef struct lo_icp_params {
  lo_gicp_params gicp;
  int init_method;
  int covariance_method;
  double max_tolerable_fitness;
  int transform_thresholding;
  double max_translation;
  double max_rotation_deg;
  double rot_sigma, trans_sigma;
  int use_fixed_covariances;
} lo_icp_params;

typedef struct lo_alignment {
  int status;
  float T_icp[16];     /* getFinalTransformation (source -> target) */
  lo_pose delta;       /* after gu::PoseInverse: what goes into the edge */
  double fitness;
  double covariance[36];
  int iterations;
  int converged;
} lo_alignment;

/* IcpLoopComputation::PerformAlignment (IcpLoopComputation.cc:420-648) on already
 * accumulated clouds.  scan1/scan2 are the UN-accumulated keyed scans (used by the
 * POINT2PLANE covariance, :632-633).  pose_21_odom = keyed_poses_[key2].between([key1]),
 * pose_21_cand = pose2.between(pose1) from the candidate message. */
int lo_perform_alignment(const lo_icp_params* p, const float* src_xyz, const float* src_normals,
                         int n_src, const float* tgt_xyz, const float* tgt_normals, int n_tgt,
                         const float* scan1_xyz, int n_scan1, const float* scan2_normals,
                         int n_scan2, const lo_pose* pose_21_odom, const lo_pose* pose_21_cand,
                         lo_alignment* out);

/* ------------------------------------------------------------------ small algebra (exposed for tests) */
void lo_mat3_inverse(const double a[9], double out[9]);   /* Eigen 3x3 cofactor inverse */
int lo_mat6_inverse(const double a[36], double out[36]);  /* partial-pivot LU */
double lo_mat6_det(const double a[36]);
void lo_sym_eig(const double* a, int n, double* evals, double* evecs); /* cyclic Jacobi, ascending */
/* Eigen::LDLT (diagonal pivoting) as used at IcpLoopComputation.cc:737-757: L (unit lower),
 * D, transpositions */
void lo_ldlt(const double a[36], double L[36], double D[6], int transp[6]);

#ifdef __cplusplus
}
#endif
#endif
