// feature_init.hip — device side of the SAC-IA initial alignment
// (IcpLoopComputation::GetSacInitialAlignment, loop_closure/src/IcpLoopComputation.cc:650-705).
//
// pcl::SampleConsensusInitialAlignment evaluates its hypotheses one after the other: draw three
// source keypoints, give each a random one of its k nearest target descriptors, estimate the
// rigid transform, score it by sum_i min(d_i^2 / 100, 1) with d_i the distance from the moved
// source keypoint i to its nearest target keypoint, keep the first strictly lowest score.
// Drawing and estimating do not depend on earlier scores, so all hypotheses are generated up
// front on the host (sac_ia.h) and scored here in one launch, one workgroup per hypothesis; the
// host then replays the "first strictly lower" selection.  The descriptor k-NN table (33-D,
// squared L2, k = 5) is one more small exhaustive kernel.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <climits>
#include <thread>

#include "lgicp_internal.h"
#include "sac_ia.h"
#include "teaser.h"

namespace lgicp {

// nn[i][0..k) = indices of the k nearest target descriptors of source descriptor i.
// One lane per source descriptor (33 floats in registers); the target descriptors stream
// through LDS in tiles of FK_TILE, two at a time: a pair of targets is stored interleaved (bin b of
// both side by side, 72 floats per pair), so that seventeen wave-uniform (broadcast) ds_read_b128
// bring both and every subtraction, multiplication and addition of the two distances is one packed
// instruction - each distance still the sum over the bins in ascending order, every operation
// rounded separately.
constexpr int FK_TILE = 64;
constexpr int FK_PAIR = 72;
typedef float fk_v2f __attribute__((ext_vector_type(2)));

//
// The targets are cut into gridDim.y ranges of `per_split` (a few hundred source descriptors
// alone would leave most of the GPU idle): every range leaves its own k best per source
// descriptor, ascending, in part_d / part_i [range][source][8]; feature_knn_merge_kernel folds
// the ranges in ascending target order with the same strict `<`, which gives exactly what one
// pass over all targets gives (lowest target index first among equal distances).
__global__ __launch_bounds__(128) void feature_knn_kernel(const float* __restrict__ src_feat,
                                                          int n_src,
                                                          const float* __restrict__ tgt_feat,
                                                          int n_tgt_all, int per_split, int k,
                                                          float* __restrict__ part_d,
                                                          int* __restrict__ part_i) {
  __shared__ __attribute__((aligned(16))) float tile[(FK_TILE / 2) * FK_PAIR];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float f[33];
#pragma unroll
  for (int b = 0; b < 33; ++b) f[b] = i < n_src ? src_feat[(size_t)i * 33 + b] : 0.f;
  float bd[8];
  int bi[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) bd[s] = __builtin_inff(), bi[s] = -1;
  auto offer = [&](float d, int j) {
    float worst = bd[0];
#pragma unroll
    for (int s = 1; s < 8; ++s)
      if (s < k) worst = bd[s];
    if (d < worst) {
      float cd = d;
      int ci = j;
#pragma unroll
      for (int s = 0; s < 8; ++s)
        if (s < k) {
          const bool lt = cd < bd[s];
          const float td = lt ? bd[s] : cd;
          const int ti = lt ? bi[s] : ci;
          bd[s] = lt ? cd : bd[s];
          bi[s] = lt ? ci : bi[s];
          cd = td;
          ci = ti;
        }
    }
  };
  const int j_begin = blockIdx.y * per_split;
  const int n_tgt = min(n_tgt_all, j_begin + per_split);
  for (int j0 = j_begin; j0 < n_tgt; j0 += FK_TILE) {
    __syncthreads();
    const int cnt = min(FK_TILE, n_tgt - j0);
    for (int e = threadIdx.x; e < cnt * 33; e += blockDim.x) {
      const int jj = e / 33, b = e % 33;
      tile[(jj >> 1) * FK_PAIR + 2 * b + (jj & 1)] = tgt_feat[(size_t)j0 * 33 + e];
    }
    __syncthreads();
    for (int jj = 0; jj < cnt; jj += 2) {
      const float4* t4 = reinterpret_cast<const float4*>(tile + (jj >> 1) * FK_PAIR);
      float t[68];
#pragma unroll
      for (int q = 0; q < 17; ++q) {
        const float4 v = t4[q];
        t[4 * q] = v.x, t[4 * q + 1] = v.y, t[4 * q + 2] = v.z, t[4 * q + 3] = v.w;
      }
      fk_v2f d = {0.f, 0.f};
#pragma unroll
      for (int b = 0; b < 33; ++b) {
        const fk_v2f fb = {f[b], f[b]}, tb = {t[2 * b], t[2 * b + 1]};
        const fk_v2f e = fb - tb;
        d += e * e;
      }
      offer(d.x, j0 + jj);
      // (the partner slot of an odd tile's last target holds stale numbers: never offered)
      if (jj + 1 < cnt) offer(d.y, j0 + jj + 1);
    }
  }
  if (i < n_src) {
    const size_t o = ((size_t)blockIdx.y * n_src + i) * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s) part_d[o + s] = bd[s], part_i[o + s] = bi[s];
  }
}

__global__ void feature_knn_merge_kernel(int n_src, int n_splits, int k,
                                         const float* __restrict__ part_d,
                                         const int* __restrict__ part_i, int* __restrict__ nn) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_src) return;
  float bd[8];
  int bi[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) bd[s] = __builtin_inff(), bi[s] = -1;
  // a range's list arrives whole (two 16-byte loads each for distances and indices, independent
  // from range to range); its unused entries are +inf and never pass the strict `<`
#pragma unroll 4
  for (int r = 0; r < n_splits; ++r) {
    const size_t o = ((size_t)r * n_src + i) * 8;
    const float4 d0 = *reinterpret_cast<const float4*>(part_d + o);
    const float4 d1 = *reinterpret_cast<const float4*>(part_d + o + 4);
    const int4 i0 = *reinterpret_cast<const int4*>(part_i + o);
    const int4 i1 = *reinterpret_cast<const int4*>(part_i + o + 4);
    const float ld[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    const int li[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (e >= k) break;
      float cd = li[e] < 0 ? __builtin_inff() : ld[e];
      int ci = li[e];
#pragma unroll
      for (int s = 0; s < 8; ++s)
        if (s < k) {
          const bool lt = cd < bd[s];
          const float td = lt ? bd[s] : cd;
          const int ti = lt ? bi[s] : ci;
          bd[s] = lt ? cd : bd[s];
          bi[s] = lt ? ci : bi[s];
          cd = td;
          ci = ti;
        }
    }
  }
  for (int s = 0; s < k; ++s) nn[(size_t)i * k + s] = bi[s];
}

// Scores hypotheses: workgroup (h, c) moves SS_QUERIES source keypoints (4 per lane) by
// hypothesis h and sweeps every target keypoint through LDS tiles, two queries per packed
// instruction; err = sum min(d2 / threshold, 1), fit = sum d2 of its chunk go to
// err_fit[(h * n_chunks + c) * 2 + {0, 1}] and are added up in chunk order by the host.
constexpr int SS_THREADS = 256;
constexpr int SS_Q = 4;
constexpr int SS_QUERIES = SS_THREADS * SS_Q;
constexpr int SS_TILE = 512;
typedef float ss_v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(SS_THREADS) void sac_score_kernel(const float* __restrict__ hyp,
                                                               const float4* __restrict__ src_kp,
                                                               int n_src,
                                                               const float4* __restrict__ tgt_kp,
                                                               int n_tgt, float threshold,
                                                               double* __restrict__ err_fit) {
  __shared__ __attribute__((aligned(16))) float lx[SS_TILE], ly[SS_TILE], lz[SS_TILE];
  __shared__ double sh[2][SS_THREADS / 64];
  const float* T = hyp + (size_t)blockIdx.x * 16;
  float t[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) t[i] = T[i];
  ss_v2f qx[SS_Q / 2], qy[SS_Q / 2], qz[SS_Q / 2];
  float best[SS_Q];
  bool have[SS_Q];
#pragma unroll
  for (int q = 0; q < SS_Q; ++q) {
    const int i = blockIdx.y * SS_QUERIES + q * SS_THREADS + threadIdx.x;
    have[q] = i < n_src;
    float x = 0.f, y = 0.f, z = 0.f;
    if (have[q]) {
      const float4 p = src_kp[i];
      x = ((t[0] * p.x + t[1] * p.y) + t[2] * p.z) + t[3];
      y = ((t[4] * p.x + t[5] * p.y) + t[6] * p.z) + t[7];
      z = ((t[8] * p.x + t[9] * p.y) + t[10] * p.z) + t[11];
    }
    qx[q / 2][q % 2] = x, qy[q / 2][q % 2] = y, qz[q / 2][q % 2] = z;
    best[q] = __builtin_inff();
  }
  for (int j0 = 0; j0 < n_tgt; j0 += SS_TILE) {
    __syncthreads();
    for (int j = threadIdx.x; j < SS_TILE; j += SS_THREADS) {
      // the tail of the last tile repeats the last target: it cannot change a minimum
      const float4 p = tgt_kp[min(j0 + j, n_tgt - 1)];
      lx[j] = p.x, ly[j] = p.y, lz[j] = p.z;
    }
    __syncthreads();
#pragma unroll 2
    for (int j = 0; j < SS_TILE; j += 4) {
      const float4 xa = *reinterpret_cast<const float4*>(lx + j);
      const float4 ya = *reinterpret_cast<const float4*>(ly + j);
      const float4 za = *reinterpret_cast<const float4*>(lz + j);
      const float tx[4] = {xa.x, xa.y, xa.z, xa.w}, ty[4] = {ya.x, ya.y, ya.z, ya.w},
                  tz[4] = {za.x, za.y, za.z, za.w};
#pragma unroll
      for (int p = 0; p < SS_Q / 2; ++p) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const ss_v2f dx = qx[p] - tx[e], dy = qy[p] - ty[e], dz = qz[p] - tz[e];
          const ss_v2f d = (dx * dx + dy * dy) + dz * dz;
          // (one v_min_f32 each; like the comparison it replaces, a NaN distance never wins)
          best[2 * p] = __builtin_fminf(best[2 * p], d.x);
          best[2 * p + 1] = __builtin_fminf(best[2 * p + 1], d.y);
        }
      }
    }
  }
  double err = 0, fit = 0;
#pragma unroll
  for (int q = 0; q < SS_Q; ++q)
    if (have[q]) {
      err += (double)((best[q] <= threshold) ? best[q] / threshold : 1.0f);  // TruncatedError
      fit += (double)best[q];
    }
  for (int off = 32; off > 0; off >>= 1) {
    err += __shfl_xor(err, off);
    fit += __shfl_xor(fit, off);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[0][wave] = err, sh[1][wave] = fit;
  __syncthreads();
  if (threadIdx.x == 0) {
    const size_t o = 2 * ((size_t)blockIdx.x * gridDim.y + blockIdx.y);
    err_fit[o] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    err_fit[o + 1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
  }
}

// The matches as TEASER++ sees them: match i * k + q pairs source keypoint i with its q-th nearest
// target keypoint in descriptor space; coordinates widened to double.
__global__ __launch_bounds__(256) void match_points_kernel(const int* __restrict__ nn, const float* __restrict__ src_xyz,
                                                           const float* __restrict__ tgt_xyz, int n_src, int k,
                                                           double* __restrict__ s, double* __restrict__ d) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= n_src * k) return;
  const int i = m / k, t = nn[m];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    s[3 * m + c] = (double)src_xyz[3 * i + c];
    d[3 * m + c] = (double)tgt_xyz[3 * t + c];
  }
}

// Scale-consistency graph of n matches (TEASER++'s inlier graph when no scale is estimated):
// bit j of row i is set when | |s_j - s_i| - |d_j - d_i| | <= beta.  A wave owns one 64 x 64 block
// of the matrix on or above the diagonal: its 64 column matches j sit in registers, one per lane,
// the row's match is read with wave-uniform addresses, and all lanes test it against their own.
// The ballot of the 64 tests is the word of that row (lane r keeps row r's word); the same test
// results, gathered per lane over the rows, are the word of row j in the mirrored block - the
// relation is symmetric and so is its arithmetic (the differences change sign, their squares do
// not), so every pair is tested once and both words leave in one store each.
__global__ __launch_bounds__(256) void consistency_graph_kernel(const double* __restrict__ src,
                                                                const double* __restrict__ dst,
                                                                int n, int words, int k, double beta,
                                                                unsigned long long* __restrict__ bits) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int w = blockIdx.x;                // column block
  const int rb = blockIdx.y * 4 + wv;      // row block
  if (rb > w || rb >= words) return;       // wave-uniform: blocks below the diagonal are mirrored
  const int i0 = rb * 64;
  const int j = w * 64 + lane;
  const bool jv = j < n;
  const int jj = jv ? j : 0;
  const double sjx = src[3 * jj], sjy = src[3 * jj + 1], sjz = src[3 * jj + 2];
  const double djx = dst[3 * jj], djy = dst[3 * jj + 1], djz = dst[3 * jj + 2];
  const int rows = min(64, n - i0);
  unsigned long long mine = 0, mirrored = 0;
  double root_a = 0;
  for (int r = 0; r < rows; ++r) {
    const int i = i0 + r;
    double e;
    // matches i = q k .. q k + k - 1 pair the same source keypoint q with its k nearest targets:
    // their source coordinates are the same numbers, and so is the source-side distance to match j
    if (r == 0 || i % k == 0) {  // wave-uniform
      const double sx = src[3 * i], sy = src[3 * i + 1], sz = src[3 * i + 2];
      double a = 0;
      e = sjx - sx, a += e * e;
      e = sjy - sy, a += e * e;
      e = sjz - sz, a += e * e;
      root_a = sqrt(a);
    }
    const double dx = dst[3 * i], dy = dst[3 * i + 1], dz = dst[3 * i + 2];
    double c = 0;
    e = djx - dx, c += e * e;
    e = djy - dy, c += e * e;
    e = djz - dz, c += e * e;
    const bool ok = jv && j != i && fabs(root_a - sqrt(c)) <= beta;
    const unsigned long long m = __ballot(ok);
    if (lane == r) mine = m;
    mirrored |= ok ? 1ull << r : 0ull;
  }
  if (lane < rows) bits[(size_t)(i0 + lane) * words + w] = mine;
  if (rb != w && jv) bits[(size_t)j * words + rb] = mirrored;
}


}  // namespace lgicp

namespace lgicp_detail {


static MatcherDev& matcher_of(lgicp_ctx* ctx, int fset) {
  if (fset < 0 || fset >= lgicp_ctx::FEATURE_SETS) fset = 0;
  if (!ctx->matcher[fset]) ctx->matcher[fset] = std::make_shared<MatcherDev>();
  return *ctx->matcher[fset];
}

// descriptor k-NN table [n_src][k] (device pointers; partial lists live in `d`)
static int launch_feature_knn(lgicp_ctx* ctx, MatcherDev& d, const float* d_sf, int n_src,
                              const float* d_tf, int n_tgt, int k, int* d_nn, hipStream_t st) {
  const int src_blocks = (n_src + 127) / 128;
  // enough target ranges for ~4 workgroups per CU, each at least one tile long (a few hundred
  // descriptors a side: the launch is as long as one workgroup's chain of tiles)
  int n_splits = std::max(1, std::min((1024 + src_blocks - 1) / src_blocks,
                                      (n_tgt + lgicp::FK_TILE - 1) / lgicp::FK_TILE));
  int per_split = (n_tgt + n_splits - 1) / n_splits;
  per_split = ((per_split + lgicp::FK_TILE - 1) / lgicp::FK_TILE) * lgicp::FK_TILE;
  n_splits = (n_tgt + per_split - 1) / per_split;
  if (d.part_d.reserve((size_t)n_splits * n_src * 8) != hipSuccess ||
      d.part_i.reserve((size_t)n_splits * n_src * 8) != hipSuccess)
    return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  float* d_pd = d.part_d.p;
  int* d_pi = d.part_i.p;
  hipLaunchKernelGGL(lgicp::feature_knn_kernel, dim3(src_blocks, n_splits), dim3(128), 0, st, d_sf,
                     n_src, d_tf, n_tgt, per_split, k, d_pd, d_pi);
  hipLaunchKernelGGL(lgicp::feature_knn_merge_kernel, dim3((n_src + 127) / 128), dim3(128), 0, st,
                     n_src, n_splits, k, d_pd, d_pi, d_nn);
  HIP_TRY(ctx, hipGetLastError());
  return LGICP_OK;
}

// SAC-IA on the keypoints/descriptors of two clouds.  score = getFitnessScore() of the
// keypoint alignment, DBL_MAX when the source has fewer than 3 keypoints or the target none.
int sac_align_device(lgicp_ctx* ctx, const CloudFeatures& src, const CloudFeatures& tgt,
                     const float guess[16], uint32_t seed, float T_out[16], double* score, int lane) {
  std::memcpy(T_out, guess, sizeof(float) * 16);
  *score = DBL_MAX;
  const int n_src = (int)src.idx.size(), n_tgt = (int)tgt.idx.size();
  if (n_src < 3 || n_tgt < 1) return LGICP_OK;
  const int k = std::min(5, n_tgt);  // setCorrespondenceRandomness(5), IcpLoopComputation.cc:697
  if (lane < 0 || lane >= lgicp_ctx::FEATURE_SETS * FEATURE_LANES || lane >= ctx->n_slots) lane = 0;
  hipStream_t st = ctx->slot[lane].stream;
  SacLane& d = ctx->sac_lanes.get()[lane];
  MatcherDev& md = d.md;
  const float* d_sf = src.d_feat;  // descriptors stay on the device with the cloud's features
  const float* d_tf = tgt.d_feat;
  if (!d_sf || !d_tf || md.nn.reserve((size_t)n_src * k) != hipSuccess ||
      d.sk.reserve((size_t)n_src) != hipSuccess || d.tk.reserve((size_t)n_tgt) != hipSuccess)
    return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  float4* d_sk = d.sk.p;
  float4* d_tk = d.tk.p;
  int* d_nn = md.nn.p;
  std::vector<float4> hs(n_src), ht(n_tgt);
  for (int i = 0; i < n_src; ++i) hs[i] = make_float4(src.xyz[3 * i], src.xyz[3 * i + 1], src.xyz[3 * i + 2], 1.f);
  for (int i = 0; i < n_tgt; ++i) ht[i] = make_float4(tgt.xyz[3 * i], tgt.xyz[3 * i + 1], tgt.xyz[3 * i + 2], 1.f);
  HIP_TRY(ctx, hipMemcpyAsync(d_sk, hs.data(), sizeof(float4) * n_src, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d_tk, ht.data(), sizeof(float4) * n_tgt, hipMemcpyHostToDevice, st));
  {
    const int rc = launch_feature_knn(ctx, md, d_sf, n_src, d_tf, n_tgt, k, d_nn, st);
    if (rc != LGICP_OK) return rc;
  }
  std::vector<int> nn((size_t)n_src * k);
  HIP_TRY(ctx, hipMemcpyAsync(nn.data(), d_nn, sizeof(int) * nn.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  const sac::Hypotheses hyp = sac::make_hypotheses(src.xyz.data(), n_src, tgt.xyz.data(), nn.data(),
                                                   k, ctx->cfg.sac_iterations, guess, seed);
  const int n_hyp = (int)(hyp.T.size() / 16);
  if (n_hyp == 0) return LGICP_OK;
  const int n_chunks = (n_src + lgicp::SS_QUERIES - 1) / lgicp::SS_QUERIES;
  if (d.hyp.reserve(hyp.T.size()) != hipSuccess || d.ef.reserve((size_t)2 * n_hyp * n_chunks) != hipSuccess)
    return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  float* d_hyp = d.hyp.p;
  double* d_ef = d.ef.p;
  HIP_TRY(ctx, hipMemcpyAsync(d_hyp, hyp.T.data(), sizeof(float) * hyp.T.size(), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(lgicp::sac_score_kernel, dim3(n_hyp, n_chunks), dim3(lgicp::SS_THREADS), 0, st,
                     d_hyp, d_sk, n_src, d_tk, n_tgt, 100.0f, d_ef);
  HIP_TRY(ctx, hipGetLastError());
  std::vector<double> efc((size_t)2 * n_hyp * n_chunks);
  HIP_TRY(ctx, hipMemcpyAsync(efc.data(), d_ef, sizeof(double) * efc.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  std::vector<double> ef((size_t)2 * n_hyp, 0.0);
  for (int h = 0; h < n_hyp; ++h)
    for (int c = 0; c < n_chunks; ++c) {
      ef[2 * h] += efc[2 * ((size_t)h * n_chunks + c)];
      ef[2 * h + 1] += efc[2 * ((size_t)h * n_chunks + c) + 1];
    }
  // the sequential rule: hypothesis 0 always, afterwards only a strictly lower error
  int best = 0;
  float lowest = (float)ef[0];
  for (int h = 1; h < n_hyp; ++h)
    if ((float)ef[2 * h] < lowest) lowest = (float)ef[2 * h], best = h;
  std::memcpy(T_out, &hyp.T[(size_t)16 * best], sizeof(float) * 16);
  *score = ef[2 * best + 1] / n_src;
  return LGICP_OK;
}

namespace {
template <class F>
void parallel_for(int n, F f) {
  const int nt = std::min(n, (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency())));
  if (nt <= 1) {
    for (int i = 0; i < n; ++i) f(i);
    return;
  }
  std::atomic<int> next{0};
  auto run = [&] {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n) return;
      f(i);
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; ++t) pool.emplace_back(run);
  run();
  for (std::thread& t : pool) t.join();
}
}  // namespace

// GetTeaserInitialAlignment from the keypoints on (IcpLoopComputation.cc:990-1081), first half:
// descriptor matches and their scale-consistency graph, built on the device from the descriptors
// and keypoints the clouds' features keep there.  Everything is enqueued on the context's stream
// and the call returns; the matched keypoints land in the job's pinned buffer.
int teaser_prepare_device(lgicp_ctx* ctx, const CloudFeatures& src, const CloudFeatures& tgt,
                          TeaserJob* job, hipStream_t st, int fset, int sub) {
  namespace tz = lgicp::teaser;
  job->n_matches = 0;
  job->src.clear();
  job->dst.clear();
  job->graph = tz::BitGraph();
  const int n_src = (int)src.idx.size(), n_tgt = (int)tgt.idx.size();
  if (n_src == 0 || n_tgt == 0) return LGICP_OK;
  const int k = std::min(5, n_tgt);  // calculateKCorrespondences(..., 5), IcpLoopComputation.cc:997
  const int n = n_src * k;
  job->n_matches = n;
  if (n <= 10) return LGICP_OK;  // IcpLoopComputation.cc:1002
  // the consistency graph is a dense n x n bit matrix: 65536 matches are 512 MiB of rows
  constexpr int kMaxMatches = LGICP_TEASER_MAX_MATCHES;
  if (n > kMaxMatches) {
    // a per-pair outcome, not a call-level error: the pair keeps its odometry guess (what the
    // reference does whenever TEASER++ does not produce a better transform,
    // IcpLoopComputation.cc:1065-1074) and the other pairs of the submit go on
    if (std::getenv("LGICP_TRACE"))
      std::fprintf(stderr, "[lgicp] teaser: %d descriptor matches exceed the %d the consistency graph "
                   "is sized for; the pair keeps its guess\n", n, kMaxMatches);
    return LGICP_OK;
  }
  if (!src.d_feat || !tgt.d_feat) return fail(ctx, LGICP_E_INVALID, "cloud features without device copies");
  if (!st) st = ctx->stream;
  // pairs prepared side by side on several streams each need a k-NN table of their own: stream 0 of
  // a feature set uses the set's matcher, the others the (otherwise idle) matchers of its SAC-IA lanes
  const bool own = sub > 0 && sub < FEATURE_LANES && fset >= 0 && fset < lgicp_ctx::FEATURE_SETS;
  MatcherDev& md = own ? ctx->sac_lanes.get()[fset * FEATURE_LANES + sub].md : matcher_of(ctx, fset);
  const tz::Params prm;
  tz::BitGraph& g = job->graph;
  g.n = n;
  g.words = (n + 63) / 64;
  const size_t n_words = (size_t)n * g.words;
  const size_t sd_bytes = sizeof(double) * 6 * (size_t)n;
  if (job->h_sd_cap < sd_bytes) {
    park_pinned_memory(job->h_sd);  // freed when the next submit begins: no device-wide wait here
    job->h_sd = nullptr, job->h_sd_cap = 0;
    const size_t want = sd_bytes + sd_bytes / 4;
    if (hipHostMalloc((void**)&job->h_sd, want, hipHostMallocDefault) != hipSuccess) {
      job->h_sd = nullptr;
      return fail(ctx, LGICP_E_NOMEM, "pinned allocation failed");
    }
    job->h_sd_cap = want;
  }
  if (md.nn.reserve((size_t)n) != hipSuccess || job->d_sd.reserve(6 * (size_t)n) != hipSuccess ||
      job->d_bits.reserve(n_words) != hipSuccess)
    return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  {
    const int rc = launch_feature_knn(ctx, md, src.d_feat, n_src, tgt.d_feat, n_tgt, k, md.nn.p, st);
    if (rc != LGICP_OK) return rc;
  }
  double* d_s = job->d_sd.p;
  double* d_d = job->d_sd.p + 3 * (size_t)n;
  hipLaunchKernelGGL(lgicp::match_points_kernel, dim3((n + 255) / 256), dim3(256), 0, st, md.nn.p, src.d_kxyz,
                     tgt.d_kxyz, n_src, k, d_s, d_d);
  hipLaunchKernelGGL(lgicp::consistency_graph_kernel, dim3(g.words, (n + 255) / 256), dim3(256), 0, st, d_s,
                     d_d, n, g.words, k, 2.0 * prm.noise_bound * std::sqrt(prm.cbar2), job->d_bits.p);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(job->h_sd, job->d_sd.p, sd_bytes, hipMemcpyDeviceToHost, st));
  // The graph stays on the device: core numbers and the clique search run there
  // (teaser_graph.hip).  Only a graph too large for that path (or reserved[4] = -1: the host
  // search, for cross-checks) is brought to the host.
  g.bits.clear();
  if (n > clique_device_max_vertices() || ctx->cfg.reserved[4] == -1) {
    g.bits.resize(n_words);
    HIP_TRY(ctx, hipMemcpyAsync(g.bits.data(), job->d_bits.p, sizeof(uint64_t) * n_words, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
  }
  return LGICP_OK;
}

// after the stream has been synchronised: the matched keypoints as the host half wants them
void teaser_prepare_collect(TeaserJob* job) {
  const size_t n = (size_t)job->graph.n;
  if (n == 0) return;
  job->src.assign(job->h_sd, job->h_sd + 3 * n);
  job->dst.assign(job->h_sd + 3 * n, job->h_sd + 6 * n);
}

// Second half, for a group of pairs at once: clique, rotation, translation and the acceptance
// test.  The clique search (teaser_graph.hip) advances all pairs of the group together on the
// device; rotation (over the chain of the clique), translation vote and error metrics are host
// work spread over threads.  T_out[i] is the guess unless the TEASER++
// transform scores at least as well over the matches.
int teaser_finish_group(lgicp_ctx* ctx, TeaserJob* const* jobs, const float* const* guesses, int count,
                        float (*T_out)[16], int* used, int* n_clique, CliqueTimes* times, hipStream_t st,
                        int fset) {
  namespace tz = lgicp::teaser;
  for (int i = 0; i < count; ++i) {
    std::memcpy(T_out[i], guesses[i], sizeof(float) * 16);
    used[i] = n_clique[i] = 0;
  }
  const tz::Params prm;
  if (!st) st = ctx->stream;
  std::vector<std::vector<int>> cliques((size_t)count);
  {
    std::vector<CliqueGraph> graphs((size_t)count, CliqueGraph{nullptr, 0, 0});
    std::vector<int> on_host;
    int on_device = 0, largest = 0;
    for (int i = 0; i < count; ++i) {
      const tz::BitGraph& g = jobs[i]->graph;
      if (g.n == 0) continue;
      if (g.bits.empty())
        graphs[i] = CliqueGraph{jobs[i]->d_bits.p, g.n, g.words}, ++on_device, largest = std::max(largest, g.n);
      else
        on_host.push_back(i);
    }
    if (on_device && !greedy_clique_group(&ctx->clique_bufs[fset + 1], ctx->cfg.device, st, graphs.data(), count,
                                          cliques.data(), times,
                                          ctx->cfg.reserved[4] == -2))
      return fail(ctx, LGICP_E_HIP, "TEASER++ initialisation: clique search over %d graphs (up to %d matches) "
                  "failed on the device (%s)", on_device, largest, hipGetErrorString(hipGetLastError()));
    // graphs the device path does not take (or reserved[4] = -1): the host search, pairs side by
    // side; a pair with ten thousand matches also spreads its own starts over threads
    parallel_for((int)on_host.size(), [&](int q) {
      const tz::BitGraph& g = jobs[on_host[q]]->graph;
      cliques[on_host[q]] = tz::greedy_clique(g, g.n >= 2048 ? 8 : 1);
    });
  }
  // rotation (GNC-TLS over the chain of the clique's consecutive matches), translation vote and the
  // acceptance test: a few thousand operations per pair, host work spread over threads
  struct Work {
    std::vector<double> cs, cd;
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
  };
  std::vector<Work> W((size_t)count);
  for (int i = 0; i < count; ++i) n_clique[i] = (int)cliques[i].size();
  parallel_for(count, [&](int i) {
    if (jobs[i]->graph.n == 0) return;
    Work& w = W[i];
    if (cliques[i].size() > 1) {
      tz::clique_points(jobs[i]->src, jobs[i]->dst, cliques[i], &w.cs, &w.cd);
      tz::clique_rotation_host(w.cs, w.cd, prm, w.R);
      tz::clique_translation(w.cs, w.cd, w.R, prm, w.t);
    }
    double Ro[9], to[3];
    for (int a = 0; a < 3; ++a) {
      for (int b = 0; b < 3; ++b) Ro[a * 3 + b] = (double)guesses[i][a * 4 + b];
      to[a] = (double)guesses[i][a * 4 + 3];
    }
    const double e_teaser = tz::error_metric(jobs[i]->src, jobs[i]->dst, w.R, w.t, 100.0);
    const double e_odom = tz::error_metric(jobs[i]->src, jobs[i]->dst, Ro, to, 100.0);
    if (e_teaser <= e_odom) {  // IcpLoopComputation.cc:1065-1068
      float* T = T_out[i];
      for (int q = 0; q < 16; ++q) T[q] = (q == 15) ? 1.f : 0.f;
      for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) T[a * 4 + b] = (float)w.R[a * 3 + b];
        T[a * 4 + 3] = (float)w.t[a];
      }
      used[i] = 1;
    }
  });
  return LGICP_OK;
}

int teaser_align_device(lgicp_ctx* ctx, const CloudFeatures& src, const CloudFeatures& tgt,
                        const float guess[16], float T_out[16], int* used, int* n_matches,
                        int* n_clique) {
  std::memcpy(T_out, guess, sizeof(float) * 16);
  *used = *n_matches = *n_clique = 0;
  TeaserJob job;
  const int rc = teaser_prepare_device(ctx, src, tgt, &job);
  *n_matches = job.n_matches;
  if (rc != LGICP_OK) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  teaser_prepare_collect(&job);
  TeaserJob* jp = &job;
  float T[1][16];
  const int rc2 = teaser_finish_group(ctx, &jp, &guess, 1, T, used, n_clique);
  if (rc2 == LGICP_OK) std::memcpy(T_out, T[0], sizeof(float) * 16);
  return rc2;
}

}  // namespace lgicp_detail
