/*
 * lamp_gicp.h — C ABI of the MI355X loop-closure verification engine.
 *
 * Drop-in boundary for LAMP's loop_closure stage 4.  The reference has no FFI for this
 * path: the stage is the C++ class lamp_loop_closure::IcpLoopComputation
 * (loop_closure/include/loop_closure/IcpLoopComputation.h:24-159) behind the abstract
 * lamp_loop_closure::LoopComputation (include/loop_closure/LoopComputation.h:22-68).  The
 * entry points below are what a sibling subclass binds instead of calling PCL: each comment
 * names the reference member it replaces.
 *
 * Conventions: plain pointers and sizes, no C++/torch types.  Matrices are ROW-MAJOR.
 * Every function returns 0 (or a non-negative count) on success and a negative
 * LGICP_E_* code on a call-level error; lgicp_last_error() gives the text.  Per-pair
 * failures are NOT errors: they come back as lgicp_result.status, one value per reject
 * reason of IcpLoopComputation::PerformAlignment, so the adapter emits the same edge set.
 *
 * Threading: one context = one submitting thread (the reference's single ROS spinner
 * thread, IcpLoopComputation.cc:189-190); add_keyframe/add_pose must not overlap submit.
 */
#ifndef LAMP_GICP_H_
#define LAMP_GICP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LGICP_ABI_VERSION 3

/* call-level error codes */
enum {
  LGICP_OK = 0,
  LGICP_E_INVALID = -1,     /* bad argument */
  LGICP_E_HIP = -2,         /* HIP runtime error (text in lgicp_last_error) */
  LGICP_E_NO_DEVICE = -3,   /* no usable gfx950 device */
  LGICP_E_UNSUPPORTED = -4, /* configuration not implemented on the device path */
  LGICP_E_NOMEM = -5
};

/* IcpLoopComputation::IcpInitMethod (IcpLoopComputation.h:135-142) */
enum {
  LGICP_INIT_IDENTITY = 0,
  LGICP_INIT_ODOMETRY = 1,
  LGICP_INIT_ODOM_ROTATION = 2,
  LGICP_INIT_FEATURES = 3,
  LGICP_INIT_TEASERPP = 4,
  LGICP_INIT_CANDIDATE = 5
};
/* TEASERPP: the scale-consistency graph over the descriptor matches of a pair is a dense bit
 * matrix (n x n / 8 bytes), sized for at most this many matches, i.e. 13,107 source keypoints; a pair
 * beyond it goes to GICP from its odometry guess and is counted in
 * lgicp_stats.teaser_oversized_pairs (the reference runs TEASER++ at any size,
 * IcpLoopComputation.cc:996-1054; with its Harris radius a keyed scan yields a few hundred to a
 * few thousand keypoints) */
#define LGICP_TEASER_MAX_MATCHES 65536
/* IcpLoopComputation::IcpCovarianceMethod (IcpLoopComputation.h:144) */
enum { LGICP_COV_POINT2POINT = 0, LGICP_COV_POINT2PLANE = 1 };

/* Inner solver of an outer GICP iteration, the part of estimateRigidTransformationBFGS
 * (lamp_utils/include/lamp_utils/gicp.hpp:217-287).
 * GAUSS_NEWTON: the objective of one outer iteration is reduced to a quadratic form in one pass
 *   over the correspondences and minimised exactly on the 6 Euler parameters; the fastest, and
 *   its answer does not move with the rounding noise of the objective.
 * BFGS: the reference's own optimiser (pcl::BFGS with the parameters of gicp.hpp:250-258, at most
 *   optimizer_iterations steps, gradient tolerance 1e-2) on the reference's own objective, every
 *   function evaluation a pass over the correspondences with T * p rounded to float as in
 *   gicp.hpp:307-310.  It stops where the reference stops, also where that is not the minimum
 *   (few outer iterations, NoProgress inside the float noise); several times slower. */
enum { LGICP_SOLVER_GAUSS_NEWTON = 0, LGICP_SOLVER_BFGS = 1 };

/* per-pair outcome: one value per `return false` of PerformAlignment */
enum {
  LGICP_PAIR_OK = 0,
  LGICP_PAIR_MISSING_SCAN = 1,   /* IcpLoopComputation.cc:438-442 */
  LGICP_PAIR_MISSING_POSE = 2,   /* :444-448 */
  LGICP_PAIR_EMPTY_CLOUD = 3,    /* :458-465 */
  LGICP_PAIR_INIT_REJECTED = 4,  /* :523-526 (SAC score too high) */
  LGICP_PAIR_NONFINITE = 5,      /* :558-559 */
  LGICP_PAIR_NOT_CONVERGED = 6,  /* :580-584 */
  LGICP_PAIR_FITNESS = 7,        /* :588-593 */
  LGICP_PAIR_TRANSFORM = 8       /* :604-614 */
};

/* Every tunable of the path.  Field comments give the reference parameter
 * (loop_closure/config/laser_parameters.yaml, IcpLoopComputation::LoadParameters
 * IcpLoopComputation.cc:58-167) or the hard-coded value it mirrors. */
typedef struct lgicp_config {
  int abi_version;            /* LGICP_ABI_VERSION */
  int device;                 /* HIP device ordinal */
  /* icp_lc/... and SetupICP (IcpLoopComputation.cc:194-203) */
  double tf_epsilon;          /* icp_lc/tf_epsilon */
  double corr_dist;           /* icp_lc/corr_dist */
  int iterations;             /* icp_lc/iterations */
  int optimizer_iterations;   /* setMaximumOptimizerIterations(50) */
  double rotation_epsilon;    /* gicp.h:128 (2e-3) */
  int k_correspondences;      /* gicp.h:121 (20) */
  double gicp_epsilon;        /* gicp.h:127 (1e-3) */
  int covariance_from_knn;    /* 0: covariances from stored normals (the path the reference
                                 takes, gicp.hpp:81-82); 1: k-NN + SVD branch (gicp.hpp:83-155) */
  int init_method;            /* icp_initialization_method */
  int covariance_method;      /* icp_covariance_calculation */
  double max_tolerable_fitness;
  int transform_thresholding; /* icp_lc/transform_thresholding */
  double max_translation;     /* icp_lc/max_translation */
  double max_rotation_deg;    /* icp_lc/max_rotation */
  int num_prev_scans;         /* sac_ia/num_prev_scans */
  int num_next_scans;         /* sac_ia/num_next_scans */
  int accumulate_source;      /* sac_ia/b_accumulate_source */
  double laser_lc_rot_sigma;  /* precision_parameters.yaml:10 */
  double laser_lc_trans_sigma;/* precision_parameters.yaml:11 */
  int use_fixed_covariances;  /* b_use_fixed_covariances */
  /* feature front end (SAC-IA / TEASER++), IcpLoopComputation.cc:90-135 */
  int sac_iterations;
  double sac_features_radius;
  double sac_fitness_score_threshold;
  double harris_threshold;
  int harris_suppression;
  double harris_radius;
  int harris_refine;
  int harris_response;
  /* device-side knobs with no counterpart in the reference */
  size_t workspace_bytes;     /* upper bound for per-batch device scratch (0 = 16 GiB) */
  size_t accumulated_cache_bytes; /* Accumulated clouds (AccumulateScans: a keyframe plus its
                                 neighbours key +- num_prev/next_scans in its frame) are kept on
                                 the device with their own search index once a submit has built
                                 them — keyed scans and poses never change once stored, so such a
                                 cloud is a function of its keys — up to this many bytes, least
                                 recently used first out (0 = 32 GiB, 1 = keep none: every submit
                                 assembles its accumulated clouds afresh, results are the same) */
  int solver;                 /* LGICP_SOLVER_* (0 = Gauss-Newton) */
  int stage_timing;           /* 1 = time every stage of a submit with events on its streams
                                 (lgicp_stats nn_ms, reduce_ms, solve_ms, cov_ms, post_ms,
                                 last_nn_kernel_ms; they stay 0 otherwise).  Off by default: the
                                 events sit between the kernels of every outer iteration and cost a
                                 lone submit 5-7 % of its time.  total_ms is always measured. */
  int reserved[8];            /* tuning/diagnostics, 0 = default: [0] queries per lane of the
                                 exhaustive 1-NN kernel (2, 4, 8); [1] its target splits;
                                 [2] 1 = use the exhaustive 1-NN kernel instead of the pruned
                                 one; [3] first-round radius of the pruned search in mm;
                                 [4] -1 = TEASER++ clique search on the host instead of the device
                                 (cross-checks), -2 = on the device, every improving start run
                                 once more on its own (exercises a fallback, same result), > 0 =
                                 diagnostics only: lookup cap in mm;
                                 [5] bit 0 = host round trip after every outer iteration, bit 1 = no
                                 limit-cycle exit from the outer loop, bit 2 = the radius sweeps of
                                 Harris / FPFH look at every point instead of skipping the
                                 stretches of a scan that lie out of reach (same results); [6] 1 = no
                                 seeding of a lookup with the previous match, 2 = seeding but no
                                 lookup certificates (every query searches in every pass; the
                                 results are the same bits either way); [7] parts of a
                                 submit in flight at once (1..4; default: up to 4, as many as
                                 streams were found whose kernels overlap) */
} lgicp_config;

/* pose_graph_msgs/LoopCandidate (pose_graph_msgs/msg/LoopCandidate.msg:1-18) */
typedef struct lgicp_candidate {
  uint64_t key_from;          /* source keyframe, key1 of PerformAlignment */
  uint64_t key_to;            /* target keyframe, key2 */
  double pose_from_q[4];      /* orientation w,x,y,z */
  double pose_from_t[3];
  double pose_to_q[4];
  double pose_to_t[3];
} lgicp_candidate;

/* what PerformAlignment returns + what CreateLoopClosureEdge (LoopComputation.cc:67-87)
 * needs to fill a pose_graph_msgs/PoseGraphEdge */
typedef struct lgicp_result {
  uint64_t key_from, key_to;
  int32_t status;             /* LGICP_PAIR_* */
  int32_t iterations;         /* nr_iterations_ of the GICP outer loop */
  int32_t converged;          /* hasConverged() */
  int32_t n_correspondences;  /* inliers of the last outer iteration */
  float T_icp[16];            /* getFinalTransformation(): source -> target, before PoseInverse */
  double delta_R[9];          /* after gu::PoseInverse (IcpLoopComputation.cc:617): edge rotation */
  double delta_t[3];          /* edge translation */
  double delta_q[4];          /* same rotation as a quaternion w,x,y,z (edge.pose.orientation) */
  double fitness;             /* getFitnessScore() -> edge.range_error (IcpLoopComputation.cc:298) */
  double covariance[36];      /* edge.covariance, order rotX,rotY,rotZ,x,y,z */
  double information[36];     /* Ap of ComputeAp_ForPoint2PlaneICP / 0.05^2 (POINT2PLANE) or
                                 inverse of covariance otherwise; zero when not computed */
  float T_init[16];           /* the guess handed to GICP (after SAC-IA when init is FEATURES) */
  double init_fitness;        /* FEATURES: sac_ia.getFitnessScore() (IcpLoopComputation.cc:701);
                                 TEASERPP: size of the inlier clique when the TEASER++ transform
                                 replaced the odometry guess, else 0; other methods: 0 */
} lgicp_result;

typedef struct lgicp_stats {
  uint64_t pairs_submitted, pairs_ok;
  uint64_t outer_iterations;    /* summed over pairs */
  uint64_t nn_launches;         /* launches of the 1-NN kernel */
  double nn_distance_evals;     /* N*M summed over launches and pairs (8 flop each) */
  double nn_ms, reduce_ms, solve_ms, cov_ms, post_ms, total_ms; /* hipEvent time, last submit; all
                                   but total_ms need lgicp_config::stage_timing */
  double last_nn_kernel_ms;     /* duration of the last 1-NN launch of the last submit (stage_timing) */
  double last_nn_evals_per_launch;
  double nn_executed_evals;     /* distances the pruned 1-NN kernel actually evaluated, last submit */
  double reduce_points;         /* source points read by the linearise/reduce kernel, summed over
                                   pairs and outer iterations, last submit */
  double reduce_inliers;        /* correspondences inside corr_dist among them */
  uint64_t reduce_launches;     /* launches of the linearise/reduce kernel, last submit */
  uint64_t cycle_exits;         /* pairs of the last submit that left the outer loop because their
                                   transformation revisited one of its last 16 values (an exit the
                                   reference does not have; reserved[5] bit 1 switches it off) */
  /* TEASERPP initialisation, last submit: the two kernels its time goes to, hipEvent durations
     summed over their launches, and the bytes of consistency-graph rows they had to read */
  double clique_ms;             /* clique_starts_group_kernel */
  double clique_row_bytes;      /* one row per start and per clique member a start tried */
  uint64_t clique_launches;
  uint64_t core_launches;
  double core_ms;               /* core_peel_group_kernel */
  double core_matrix_bytes;     /* the bit matrix of every graph, once */
  uint64_t teaser_oversized_pairs; /* TEASERPP, last submit: pairs whose descriptor matches (5 per source
                                   keypoint) exceeded LGICP_TEASER_MAX_MATCHES; such a pair keeps its
                                   odometry guess, as a pair whose TEASER++ transform is not better
                                   than the guess does (IcpLoopComputation.cc:1065-1074) */
} lgicp_stats;

typedef struct lgicp_ctx lgicp_ctx;

/* values of the reference's "base" (base_station != 0) or "robot" parameter block */
void lgicp_default_config(lgicp_config* cfg, int base_station);

/* IcpLoopComputation::Initialize / LoadParameters / SetupICP */
int lgicp_create(const lgicp_config* cfg, lgicp_ctx** out);
void lgicp_destroy(lgicp_ctx* ctx);
const char* lgicp_last_error(const lgicp_ctx* ctx); /* ctx may be NULL: last create error */

/* IcpLoopComputation::KeyedScanCallback (IcpLoopComputation.cc:377-392): first insert of a
 * key wins.  xyz / normals point at the first float of point 0; stride_bytes is the distance
 * between points (48 for pcl::PointXYZINormal, with normals = xyz + 4 floats).  normals may
 * be NULL (all-zero normals).  Returns 1 if stored, 0 if the key already had a scan. */
int lgicp_add_keyframe(lgicp_ctx* ctx, uint64_t key, const float* xyz, const float* normals,
                       size_t n_points, size_t stride_bytes);
/* The same for `count` keyed scans at once (one device allocation, one upload and one pass of
 * every registration kernel for the whole batch; a backlog of keyed scans, or a data set, is
 * registered several times faster this way than scan by scan).  xyz[k] / normals[k] / n_points[k]
 * describe scan k as above; normals, or single entries of it, may be NULL.  Keys already stored,
 * or repeated inside the call, are skipped (first wins).  Returns the number of scans stored.
 * What registration derives on the device is what the reference builds inside every alignment:
 * the search structure of setInputTarget (lamp_utils/include/lamp_utils/gicp.h:175-209) and the
 * per-point covariances of computeCovariances (gicp.hpp:422-432), plus NormalizePCloud's
 * constants (PointCloudUtils.cc:39-59).  The device work is only enqueued: lgicp_submit waits for
 * it by itself, lgicp_wait_registered does so explicitly. */
int lgicp_add_keyframes(lgicp_ctx* ctx, size_t count, const uint64_t* keys, const float* const* xyz,
                        const float* const* normals, const size_t* n_points, size_t stride_bytes);
/* returns once every keyframe registered so far is resident and indexed on the device */
int lgicp_wait_registered(lgicp_ctx* ctx);
/* IcpLoopComputation::KeyedPoseCallback (IcpLoopComputation.cc:394-418): first wins. */
int lgicp_add_pose(lgicp_ctx* ctx, uint64_t key, const double q_wxyz[4], const double t[3]);
int lgicp_has_keyframe(const lgicp_ctx* ctx, uint64_t key);
int lgicp_has_pose(const lgicp_ctx* ctx, uint64_t key);

/* The batch body of IcpLoopComputation::ComputeTransforms (IcpLoopComputation.cc:243-367):
 * PerformAlignment for every candidate, all pairs in flight on the device at once (what the
 * reference's ThreadPool, ThreadPool.h:17-115, does with host threads).  out[n] is caller
 * allocated; out[i] belongs to cand[i].  Blocking, like the reference's future.wait() loop. */
int lgicp_submit(lgicp_ctx* ctx, const lgicp_candidate* cand, size_t n, lgicp_result* out);
/* The same, and the statistics of THIS submit in *submit_stats (may be NULL): with several submits
 * in flight (lgicp_set_submit_lanes) lgicp_get_stats' "last submit" fields belong to whichever
 * submit ended last.  Cumulative fields (pairs_submitted, outer_iterations, nn_launches,
 * nn_distance_evals) hold this submit's share. */
int lgicp_submit_stats(lgicp_ctx* ctx, const lgicp_candidate* cand, size_t n, lgicp_result* out,
                       lgicp_stats* submit_stats);
/* How many lgicp_submit calls of different host threads run side by side on the context (1..8;
 * default 1: a submit has the whole context, a second caller waits).  The reference spreads the
 * candidates of a tick over a pool of host threads (IcpLoopComputation.cc:327-365, ThreadPool.h);
 * here a submit already carries all its pairs on the device at once, and what a second submit in
 * flight adds is that the last outer iterations of one batch -- a handful of pairs, latency-bound
 * -- run beside the full-size launches of the others.  With n lanes every submit works with a
 * share of the context's streams and batch buffers (one stream each from 4 lanes on; measured on
 * 256-pair batches of 16k-point keyframes: 4.5 ms per batch with one submit at a time, 3.3 ms with
 * four in flight, 3.1 ms with six).  A submit that finds no other submit in flight is spread over
 * as many streams as with one lane, so a caller that is alone loses nothing by the setting.
 * lgicp_submit may then be called from n threads at once (more callers wait their turn); registration (lgicp_add_keyframe(s), lgicp_add_pose,
 * lgicp_wait_registered) and lgicp_align wait for the submits in flight and hold new ones off
 * while they run.  A context with a feature-based initialisation (FEATURES, TEASERPP) has at most
 * two lanes (each works with its own set of feature buffers and four streams; 512-pair TEASERPP
 * batches: 0.65 s per batch in turn, 0.50 s with two in flight).  The single-operation entry points below (lgicp_nn_search,
 * lgicp_fpfh, ...) use the context's first stream and are not to be called while submits are in
 * flight.  Every lane holds its own batch buffers (sized by its largest batch so far, at most
 * workspace_bytes of device memory each).  lgicp_get_stats' "last submit" fields describe
 * whichever submit ended last.  Returns the number of lanes now in effect, or a negative LGICP_E_* code. */
int lgicp_set_submit_lanes(lgicp_ctx* ctx, int lanes);

int lgicp_get_stats(const lgicp_ctx* ctx, lgicp_stats* out);

/* ------------------------------------------------------------------ kernel-level entry
 * points (host buffers in, host buffers out; used by the parity tests and by callers that
 * want one stage only).  Points are packed xyz floats (stride 3). */

/* one pass of the correspondence search of gicp.hpp:463-498: q = T * src (float), 1-NN in
 * tgt, squared distance; idx = -1 where no neighbour exists.  n_splits > 1 forces the
 * target to be split over that many workgroups per query tile (0 = automatic). */
int lgicp_nn_search(lgicp_ctx* ctx, const float T[16], const float* src_xyz, size_t n_src,
                    const float* tgt_xyz, size_t n_tgt, int32_t* idx, float* d2, int n_splits);
/* CalculateCovarianceFromNormals (gicp_utils.h:50-62): cov9 is [n][9] doubles */
int lgicp_covariances_from_normals(lgicp_ctx* ctx, const float* normals, size_t n, double* cov9);
/* computeCovariances k-NN branch (gicp.hpp:83-155) */
int lgicp_covariances_knn(lgicp_ctx* ctx, const float* xyz, size_t n, int k, double eps,
                          double* cov9);
/* MultithreadedGICP align (gicp.hpp:405-617) for one pair with an explicit guess */
int lgicp_align(lgicp_ctx* ctx, const float* src_xyz, const float* src_normals, size_t n_src,
                const float* tgt_xyz, const float* tgt_normals, size_t n_tgt,
                const float guess[16], float final_T[16], int* converged, int* iterations,
                double* fitness);
/* NormalizePCloud + ComputeAp_ForPoint2PlaneICP (PointCloudUtils.cc:39-59,115-163):
 * corr[i] >= n_ref contributes a zero normal.  Ap is 6x6 row-major. */
int lgicp_compute_ap(lgicp_ctx* ctx, const float* query_xyz, size_t n_query,
                     const float* ref_normals, size_t n_ref, const int32_t* corr, size_t n_corr,
                     const float T[16], int normalize_query, double Ap[36]);

/* lamp_utils::ComputeFeatures (lamp_utils/src/PointCloudUtils.cc:98-113): FPFH (PCL
 * FPFHSignature33, 11 bins each of f1 | f2 | f3) at the keypoints, search surface = the whole
 * cloud with its normals, radius search.  out is [n_key][33] floats. */
int lgicp_fpfh(lgicp_ctx* ctx, const float* surface_xyz, const float* surface_normals, size_t n,
               const float* keypoints_xyz, size_t n_key, double radius, float* out);

/* lamp_utils::ComputeKeypoints (PointCloudUtils.cc:61-96): Harris3D keypoints with the context's
 * harris_* parameters.  normals != NULL is the overload the path uses (:78-96); normals == NULL is
 * the overload without normals (:61-76), where pcl::HarrisKeypoint3D estimates them itself over a
 * radius search with the Harris radius.  Writes up to `capacity` indices (ascending) and the total
 * count. */
int lgicp_harris_keypoints(lgicp_ctx* ctx, const float* xyz, const float* normals, size_t n,
                           int32_t* idx_out, size_t capacity, size_t* count);
/* IcpLoopComputation::GetSacInitialAlignment (IcpLoopComputation.cc:650-705): Harris keypoints
 * and FPFH descriptors of both clouds on the device, SAC-IA (sac_iterations, 5 candidate
 * correspondences; hypotheses drawn on the host, scored on the device).  guess is scored as the first hypothesis when it is not the
 * identity.  fitness_score = getFitnessScore() of the keypoint alignment (DBL_MAX when a cloud
 * yields fewer than 3 / 1 keypoints). */
int lgicp_sac_initial_alignment(lgicp_ctx* ctx, const float* src_xyz, const float* src_normals,
                                size_t n_src, const float* tgt_xyz, const float* tgt_normals,
                                size_t n_tgt, const float guess[16], float T_out[16],
                                double* fitness_score);

/* lamp_utils::ComputeNormals, search_method 0 (PointCloudUtils.h:38-62; called for every keyed
 * scan at factor_handlers/src/PoseGraphHandler.cc:188-196 and, through ExtractNormals, whenever a
 * cloud arrives without normals, PointCloudUtils.cc:17-35): pcl::NormalEstimation over the k
 * nearest neighbours (1..32; the reference uses 10), viewpoint at the origin.  normals is
 * [n][3]; curvature (may be NULL) is [n]. */
int lgicp_compute_normals(lgicp_ctx* ctx, const float* xyz, size_t n, int k, float* normals,
                          float* curvature);

/* lamp_utils::ComputeIcpObservability (PointCloudUtils.cc:165-196): ascending eigenvalues of the
 * translational 3x3 block of Ap computed on the normalised cloud against its own (extracted)
 * normals. */
int lgicp_icp_observability(lgicp_ctx* ctx, const float* xyz, const float* normals, size_t n,
                            double eigenvalues[3]);

/* pcl::VoxelGrid with a cubic leaf, all fields averaged — the filter inside
 * LampPcldFilter::AdaptiveGridFilter (lamp_utils/src/LampPcldFilter.cc:46-50) and the
 * evaluation driver's grid filters (loop_closure/src/eval_loop_computation_test.cc:151-154,
 * 244-247).  One output point per occupied cell in ascending cell index: the float mean of its
 * points and, when normals/out_normals are given, the normalised sum of their normals.
 * Non-finite points are skipped.  When the grid would need more than INT_MAX cells the input
 * comes back unchanged, as PCL does.  out_* hold up to n points; *n_out receives the count. */
int lgicp_voxel_grid(lgicp_ctx* ctx, const float* xyz, const float* normals, size_t n,
                     double leaf_size, float* out_xyz, float* out_normals, size_t* n_out);
/* pcl::RandomSample (LampPcldFilter.cc:26-33, eval_loop_computation_test.cc:193-196,234-237):
 * exactly min(n_keep, n) points, input order kept.  PCL seeds rand() with the wall clock; here
 * the draw is a function of `seed` alone. */
int lgicp_random_sample(lgicp_ctx* ctx, const float* xyz, const float* normals, size_t n,
                        size_t n_keep, uint32_t seed, float* out_xyz, float* out_normals,
                        size_t* n_out);

/* IcpLoopComputation::GetTeaserInitialAlignment (IcpLoopComputation.cc:949-1081): the same
 * keypoint/descriptor front end, every source descriptor matched with its 5 nearest target
 * descriptors, TEASER++ (noise bound 0.5, GNC-TLS rotation, greedy max-clique inlier selection)
 * on the matches.  T_out is the guess unless there are more than 10 matches and the TEASER++
 * transform's truncated error over them is not larger than the guess's; *used_teaser tells
 * which.  n_matches / n_clique (may be NULL) report the sizes of the match set and of the
 * inlier clique. */
int lgicp_teaser_initial_alignment(lgicp_ctx* ctx, const float* src_xyz, const float* src_normals,
                                   size_t n_src, const float* tgt_xyz, const float* tgt_normals,
                                   size_t n_tgt, const float guess[16], float T_out[16],
                                   int32_t* used_teaser, int32_t* n_matches, int32_t* n_clique);

/* The inlier selection inside TEASER++ as GetTeaserInitialAlignment configures it
 * (IcpLoopComputation.cc:1011: inlier_selection_mode PMC_HEU): core numbers of the consistency
 * graph, then the greedy clique heuristic over the vertices in core order.  adjacency: n rows of
 * (n + 63) / 64 64-bit words, bit j of row i set when matches i and j are consistent (symmetric,
 * no self loops, bits beyond n clear).  clique_out (room for n entries) receives the clique's
 * vertices in ascending order.  At most 65536 vertices.  The step lgicp_submit and
 * lgicp_teaser_initial_alignment run on the graphs they build themselves, exposed for graphs a
 * caller brings. */
int lgicp_teaser_clique(lgicp_ctx* ctx, const uint64_t* adjacency, size_t n, int32_t* clique_out,
                        int32_t* n_clique);

/* No counterpart in the reference.  lgicp_nn_search for callers that cannot allocate device
 * memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc):
 * the same search with the same bits out, but every buffer is the caller's DEVICE memory
 * (d_src_xyz / d_tgt_xyz packed xyz floats, d_idx[n_src], d_d2[n_src]) and all working memory is
 * one static device array of the library (about 2.6 MiB per GPU, held until the process ends), so
 * the call performs no hipMalloc / hipFree.  Each cloud may hold at most LGICP_STATIC_MAX_POINTS
 * points; larger inputs are rejected with LGICP_E_INVALID.  T is read on the host at call time.
 * hip_stream = NULL: runs on the context's stream and returns when the results are in d_idx / d_d2;
 * otherwise a hipStream_t on the context's device: the work is enqueued there and the call returns
 * at once.  All calls on one GPU share the static array: calls that overlap on one GPU race. */
#define LGICP_STATIC_MAX_POINTS 65536
int lgicp_nn_search_static(lgicp_ctx* ctx, const float T[16], const float* d_src_xyz, size_t n_src,
                           const float* d_tgt_xyz, size_t n_tgt, int32_t* d_idx, float* d_d2,
                           void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* LAMP_GICP_H_ */
