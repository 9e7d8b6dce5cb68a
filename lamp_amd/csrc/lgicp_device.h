// lgicp_device.h — structures shared by the HIP kernels and the host driver of the
// MI355X loop-closure verification engine.  gfx950 only (64-wide wavefronts).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lgicp {

// Packed (squared distance, target index) of a 1-NN result.  d2 >= 0, so its float bits
// order like the value; min over the 64-bit key = smallest distance, lowest index on ties.
typedef unsigned long long nnkey_t;
static constexpr nnkey_t NNKEY_NONE = 0xFFFFFFFFFFFFFFFFull;

// coordinate of the points that pad a target cloud up to a whole LDS tile: far enough that
// its squared distance (~3e36) never beats a real point, small enough not to overflow
static constexpr float PAD_COORD = 1.0e18f;

static constexpr int NN_THREADS = 256;
static constexpr int NN_TILE = 1024;  // target points per LDS tile (12 KiB as SoA floats)
static constexpr int NN_GROUP = 8;    // targets whose minimum is tracked together

// spatial index of a cloud: points in k-d partition order, in tiles of SI_TILE with one box per tile
static constexpr int SI_TILE = 32;
static constexpr int NNP_THREADS = 64;   // one wave of 64 queries per workgroup: waves finish independently
static constexpr int NN_COUNTER_SLOTS = 4096;  // the evaluated-distance counter is spread over these

// number of f64 statistics per pair and outer iteration: 10 x 6 second moments
// sum(u_k u_l M), 4 x 3 first moments sum(u_k M r0), sum(r0^T M r0), correspondence count
static constexpr int NSTAT = 74;
static constexpr int RED_THREADS = 256;   // linearise/reduce: four waves per slab of up to 1024 points,
                                          // each taking every fourth round of 64; their sums meet in
                                          // a fixed order.  (One wave per slab, 16 rounds in a row,
                                          // made a lone pair's reduction 20 us of its ~59 us
                                          // iteration; four waves: 46 us per iteration, same rate
                                          // for full-size launches.)
static constexpr int POST_THREADS = 256;  // post-alignment reductions
static constexpr int RED_MAX_CHUNKS = 16;
static constexpr int RED_CHUNK_POINTS = 1024;  // source points per slab while a cloud has at most
                                               // RED_MAX_CHUNKS * RED_CHUNK_POINTS points

// fitness sum, count, non-finite flag, 21 unique entries of Ap
static constexpr int NPOST = 24;

struct CloudDev {
  const float4* xyz;  // n_pad points (x, y, z, 1), tail = PAD_COORD
  const float4* nrm;  // n normals (x, y, z, 0) as stored with the points
  const float4* nrm_x;  // lamp_utils::ExtractNormals of the cloud: nrm, or k-NN normals when the
                        // first stored normal is all zero (PointCloudUtils.cc:17-35)
  const double* cov;  // n x 6: c00 c01 c02 c11 c12 c22
  const float4* rec;  // what the reduction gathers per match.  compact == 0: n records of 64 bytes,
                      // the point (x, y, z, 1), then the same six doubles.  compact == 1: n records
                      // of 32 bytes, the point (x, y, z, 1), then cnrm's entry of the point
  const float4* cnrm; // compact == 1: the stored normal (x, y, z) and a flag w: 0 = the covariance
                      // of the point is 1 - 0.999 n n^T / |n|^2 to rounding (rebuilt where it is
                      // needed), 1 = one of the reference's special cases (zero or non-finite
                      // normal, clamped z): take cov
  int compact;
  int n;
  int n_pad;
  // spatial index (spatial_index.hip)
  const float4* sxyz;  // n_tiles tiles of SI_TILE points in k-d partition order, each tile stored as
                       // x[SI_TILE] y[SI_TILE] z[SI_TILE] original-index[SI_TILE]; unused
                       // slots hold +inf coordinates and index -1
  const float4* tmin;  // n_tiles box corners, then the boxes of the ceil(n_tiles / 64) blocks of 64
  const float4* tmax;  // consecutive tiles (spatial_index.hip)
  const int* qorder;   // n_q original indices in visiting order, -1 in the padding
  int n_tiles;
  int n_q;             // n rounded up to a multiple of 64
  // NormalizePCloud constants (used only for the un-accumulated source keyframe)
  float centroid[3];
  float factor;
};

// Covariance of a point from its stored normal when none of the special branches of
// PCLTwoPlaneVectorsFromNormal (lamp_utils/include/lamp_utils/gicp_utils.h:28-48) applies: the
// reference builds C = 0.001 n n^T + v1 v1^T + v2 v2^T from the normalised normal n and two unit
// vectors that complete it to an orthonormal basis, i.e. C = 1 - 0.999 n n^T; written with the
// unnormalised normal, C = 1 - (0.999 / |n|^2) n n^T, which costs one division and agrees with the
// reference's construction to 6e-16 absolute (both carry a few roundings of entries <= 1).
__device__ __forceinline__ void cov_of_regular_normal(const float4 nf, double c[6]) {
  const double nx = nf.x, ny = nf.y, nz = nf.z;
  const double k = 0.999 / (nx * nx + ny * ny + nz * nz);
  const double kx = k * nx, ky = k * ny, kz = k * nz;
  c[0] = 1.0 - kx * nx;
  c[1] = -(kx * ny);
  c[2] = -(kx * nz);
  c[3] = 1.0 - ky * ny;
  c[4] = -(ky * nz);
  c[5] = 1.0 - kz * nz;
}

// 1 when the reference's construction takes a special branch for this normal (zero or non-finite
// length: nothing to normalise; z component clamped to 1e-7 before or after normalisation): the
// closed form above does not describe those, their covariance is read from memory instead.
__device__ __forceinline__ float normal_is_special(const float4 nf) {
  const double nx = nf.x, ny = nf.y, nz = nf.z;
  const double z = nx * nx + ny * ny + nz * nz;
  if (!(z > 0) || !(z < 1.0e300)) return 1.f;
  const double nzn = nz / sqrt(z);
  return (fabs(nz) < 0.0000001 || fabs(nzn) < 0.0000001) ? 1.f : 0.f;
}

struct PairDev {
  int src_cloud;    // accumulated source
  int tgt_cloud;    // accumulated target
  int scan1_cloud;  // un-accumulated source keyframe (query of the POINT2PLANE covariance)
  int scan2_cloud;  // un-accumulated target keyframe (normals of the POINT2PLANE covariance)
  float4* out;      // guess * source, n_src points
  nnkey_t* keys;    // n_src packed 1-NN results
  float4* cert;     // optional, n_src: where each query stood at its last full lookup (x, y, z) and
                    // a lower bound (w) of its distance from there to every target but its match
  // BFGS back end only: one record per source point and outer iteration (gicp_bfgs.hip)
  float4* bf_a;     // guess * source (x, y, z), matched target x
  float2* bf_b;     // matched target y, z
  double2* bf_m;    // 3 per point: M00 M01 | M02 M11 | M12 M22; all zero without a correspondence
  float guess[16];
  int n_src;
  int n_tgt;
};

struct PairState {
  float T[16];       // transformation_
  float Tprev[16];   // previous_transformation_
  float Tfinal[16];  // previous_ * guess
  unsigned long long Thash[16];  // hashes of the lookup transformations of the last iterations
  double delta;
  int iterations;
  int converged;
  int active;  // still inside the outer loop
  int failed;  // optimiser "exception" (fewer than 4 correspondences)
  int m;       // correspondences of the last outer iteration
  int cycle;   // 1: left the loop through the limit-cycle exit
  double m_sum;  // correspondences summed over the outer iterations (statistics only)
};

struct SolveParams {
  double corr_dist2;
  double rotation_epsilon;
  double tf_epsilon;
  int max_iterations;
  int max_inner_iterations;
  int cycle_exit;  // 1: stop a pair whose transformation revisits one of its last 16 values
  int pad;
};

// ---- launchers (each enqueues on `stream`, returns the hipError_t of the launch) ----------

// dst/src 16-byte aligned, both readable/writable up to the next multiple of 16 bytes
hipError_t launch_copy16(void* dst, const void* src, size_t bytes, hipStream_t stream);
hipError_t launch_pad_fill(float4* xyz, int n, int n_pad, hipStream_t stream);
// one idle wave that occupies `stream` for the given time
hipError_t launch_delay(double microseconds, hipStream_t stream);
hipError_t launch_cov_from_normals(const float4* nrm, int n, double* cov6, hipStream_t stream);
hipError_t launch_pack_records(const float4* xyz, const double* cov6, int n, float4* rec,
                               hipStream_t stream);

struct SegmentDev {  // one keyframe appended to an accumulated cloud
  const float4* src_xyz;
  const float4* src_nrm;
  float4* dst_xyz;
  float4* dst_nrm;
  const double* src_cov;  // optional: the keyframe's covariances-from-normals, copied along
  double* dst_cov;
  const float4* src_cnrm;  // with src_cov: the keyframe's flagged normals, copied along, and the
  float4* dst_cnrm;        // 32-byte records (moved point + flagged normal) of the new cloud
  float4* dst_rec;
  double T[12];  // row-major 3x4, applied in double then cast (transformPointCloud<double>)
  int n;
  int identity;  // 1: plain copy (the key's own scan)
};
hipError_t launch_assemble(const SegmentDev* segs, int n_segs, int max_n, hipStream_t stream);

hipError_t launch_guess_transform(const CloudDev* clouds, const PairDev* pairs, int n_pairs,
                                  int max_n, hipStream_t stream);
hipError_t launch_normalize_constants(CloudDev* clouds, const int* cloud_ids, int n_ids,
                                      hipStream_t stream);

// bmap entry: x = pair, y = query tile, z = split, w = number of splits
hipError_t launch_nn_search(const CloudDev* clouds, const PairDev* pairs, const PairState* st,
                            const int4* bmap, int n_blocks, bool final_pass, bool any_split,
                            int variant, hipStream_t stream);
int nn_queries_per_block(int variant);

// pruned exact 1-NN (knn_pruned.hip).  bmap entry: x = pair, y = group of NNP_THREADS queries.
// cap = squared radius beyond which matches are not needed (+inf: exact for every query),
// r0sq = squared radius of the first round; use_seed: PairDev::keys holds the previous pass's
// result (or NNKEY_NONE) and bounds the search; distances evaluated are added into
// counters[0 .. NN_COUNTER_SLOTS) (sum them)
hipError_t launch_nn_pruned(const CloudDev* clouds, const PairDev* pairs, const PairState* st,
                            const int4* bmap, int n_blocks, bool final_pass, float cap, float r0sq,
                            bool use_seed, unsigned long long* counters, hipStream_t stream);

// One segment of a cloud's spatial index: n points of `xyz + point_off` visited in `order`
// (a permutation of 0..n-1) fill tiles tile_off.. of the cloud's index arrays.
struct IndexSegDev {
  const float4* xyz;   // the cloud's points in original order
  const int* order;    // visiting order of this segment's own points
  float4* sxyz;
  float4* tmin;        // room for the cloud's tiles plus one entry per block of 64 tiles
  float4* tmax;
  int* qorder;
  int point_off;       // first original index of the segment inside the cloud
  int tile_off;
  int n;
  int last;            // 1: also pad qorder up to n_q after this segment
  int n_q;
};
hipError_t launch_index_fill(const IndexSegDev* segs, int n_segs, int max_n, hipStream_t stream);
hipError_t launch_keys_fill(const PairDev* pairs, const PairState* st, int n_pairs, int max_n,
                            bool final_pass, hipStream_t stream);
hipError_t launch_keys_unpack(const nnkey_t* keys, int n, int32_t* idx, float* d2,
                              hipStream_t stream);

// bmap entry: x = pair, y = chunk, z = number of chunks
// compact: the clouds carry flagged normals and 32-byte records (CloudDev::compact == 1, all of them)
hipError_t launch_gicp_reduce(const CloudDev* clouds, const PairDev* pairs, const PairState* st,
                              const int4* bmap, int n_blocks, double corr_dist2, double* partials,
                              bool compact, hipStream_t stream);
// n_done: pinned host memory, 1 + n_pairs ints: entry 1 + p receives `iter` (>= 1) when pair p
// leaves the loop, entry 0 receives `iter` when every workgroup of the launch has finished;
// ticket: one int of device memory that is zero between launches
hipError_t launch_gicp_solve(const PairDev* pairs, PairState* st, int n_pairs,
                             const int* n_chunks, const double* partials, SolveParams sp,
                             int* n_done, int iter, int* ticket, hipStream_t stream);
// BFGS back end (gicp_bfgs.hip): records + correspondence counts of one outer iteration (same block
// map as launch_gicp_reduce; counts land in partials[(pair, chunk)][0]), then the inner solve,
// applyState and the convergence test, one workgroup per pair
hipError_t launch_bfgs_prepare(const CloudDev* clouds, const PairDev* pairs, const PairState* st,
                               const int4* bmap, int n_blocks, double corr_dist2, double* partials,
                               bool compact, hipStream_t stream);
hipError_t launch_gicp_bfgs(const PairDev* pairs, PairState* st, int n_pairs, const int* n_chunks,
                            const double* partials, SolveParams sp, int* n_done, int iter,
                            int* ticket, hipStream_t stream);
hipError_t launch_post_reduce(const CloudDev* clouds, const PairDev* pairs, const PairState* st,
                              const int4* bmap, int n_blocks, double* partials, bool point2point,
                              hipStream_t stream);

hipError_t launch_cov_knn(const float4* xyz, int n, int n_pad, int k, double eps, double* cov6,
                          hipStream_t stream);
// pcl::NormalEstimation with a k-NN search: normals_curv[n] = (nx, ny, nz, curvature)
hipError_t launch_normals_knn(const float4* xyz, int n, int n_pad, int k, float4* normals_curv,
                              hipStream_t stream);

// pcl::NormalEstimation with a radius search (what pcl::HarrisKeypoint3D runs when it is given no
// normals): normals_curv[n] = (nx, ny, nz, curvature), NaN with fewer than 3 neighbours
hipError_t launch_normals_radius(const float4* xyz, int n, float r2, float4* normals_curv,
                                 hipStream_t stream);

// FPFH at n_key keypoints over a surface of n points (radius^2 = r2); need[n], spfh[n*33]
// are scratch, out is [n_key][33]
// boxes: the surface's chunk boxes (launch_chunk_boxes) or null
hipError_t launch_fpfh(const float4* surf, const float4* nrm, int n, const float4* keys, int n_key,
                       float r2, unsigned char* need, float* spfh, float* out, const float4* boxes,
                       hipStream_t stream);

// boxes[2 c], boxes[2 c + 1]: smallest and largest finite coordinates of points 64 c .. 64 c + 63 of a
// cloud in its own order (2 * ceil(n / 64) entries): what the radius sweeps of the feature front end
// use to skip the stretches of a scan that lie out of reach
hipError_t launch_chunk_boxes(const float4* xyz, int n, float4* boxes, hipStream_t stream);

#ifdef __HIPCC__
// Chunks of the tile that starts at `base` (bit k: points base + 64 k ..) whose box lies within the
// radius of (px, py, pz); lanes 0 .. tile / 64 - 1 test one chunk each.  Without boxes: every chunk.
__device__ __forceinline__ unsigned chunks_in_reach(const float4* __restrict__ boxes, int n, int base,
                                                    int tile_chunks, int lane, float px, float py,
                                                    float pz, float r2) {
  if (!boxes) return 0xFFFFFFFFu;
  const int ch = (base >> 6) + lane;
  bool in = false;
  if (lane < tile_chunks && ch * 64 < n) {
    const float4 b0 = boxes[2 * ch], b1 = boxes[2 * ch + 1];
    const float gx = __builtin_fmaxf(0.f, __builtin_fmaxf(b0.x - px, px - b1.x));
    const float gy = __builtin_fmaxf(0.f, __builtin_fmaxf(b0.y - py, py - b1.y));
    const float gz = __builtin_fmaxf(0.f, __builtin_fmaxf(b0.z - pz, pz - b1.z));
    in = (gx * gx + gy * gy) + gz * gz < r2;
  }
  return (unsigned)__ballot(in);
}
#endif

// Harris3D response + non-maxima suppression (radius^2 = r2); response[n], is_key[n]; boxes: the
// cloud's chunk boxes (launch_chunk_boxes) or null (every chunk is swept; same result)
hipError_t launch_harris(const float4* xyz, const float4* nrm, int n, float r2, float threshold,
                         float* response, unsigned char* is_key, const float4* boxes,
                         hipStream_t stream);

}  // namespace lgicp
