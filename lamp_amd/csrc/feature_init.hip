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
#include <cfloat>

#include "lgicp_internal.h"
#include "sac_ia.h"
#include "teaser.h"

namespace lgicp {

// nn[i][0..k) = indices of the k nearest target descriptors of source descriptor i.
// One lane per source descriptor (33 floats in registers); the target descriptors stream
// through LDS in tiles of FK_TILE, padded to 36 floats so that a descriptor is read back with
// nine wave-uniform (broadcast) ds_read_b128.
constexpr int FK_TILE = 64;
constexpr int FK_STRIDE = 36;

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
  __shared__ __attribute__((aligned(16))) float tile[FK_TILE * FK_STRIDE];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float f[36];
#pragma unroll
  for (int b = 0; b < 36; ++b) f[b] = (b < 33 && i < n_src) ? src_feat[(size_t)i * 33 + b] : 0.f;
  float bd[8];
  int bi[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) bd[s] = __builtin_inff(), bi[s] = -1;
  const int j_begin = blockIdx.y * per_split;
  const int n_tgt = min(n_tgt_all, j_begin + per_split);
  for (int j0 = j_begin; j0 < n_tgt; j0 += FK_TILE) {
    __syncthreads();
    const int cnt = min(FK_TILE, n_tgt - j0);
    for (int e = threadIdx.x; e < cnt * 33; e += blockDim.x)
      tile[(e / 33) * FK_STRIDE + e % 33] = tgt_feat[(size_t)j0 * 33 + e];
    __syncthreads();
    for (int jj = 0; jj < cnt; ++jj) {
      const float4* t4 = reinterpret_cast<const float4*>(tile + jj * FK_STRIDE);
      float t[36];
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const float4 v = t4[q];
        t[4 * q] = v.x, t[4 * q + 1] = v.y, t[4 * q + 2] = v.z, t[4 * q + 3] = v.w;
      }
      float d = 0;
#pragma unroll
      for (int b = 0; b < 33; ++b) {
        const float e = f[b] - t[b];
        d += e * e;
      }
      const int j = j0 + jj;
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
  for (int r = 0; r < n_splits; ++r) {
    const size_t o = ((size_t)r * n_src + i) * 8;
    for (int e = 0; e < k; ++e) {
      float cd = part_d[o + e];
      int ci = part_i[o + e];
      if (ci < 0) break;
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
          best[2 * p] = d.x < best[2 * p] ? d.x : best[2 * p];
          best[2 * p + 1] = d.y < best[2 * p + 1] ? d.y : best[2 * p + 1];
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

// Scale-consistency graph of n matches (TEASER++'s inlier graph when no scale is estimated):
// bit j of row i is set when | |s_j - s_i| - |d_j - d_i| | <= beta.  One lane per 64-bit word.
__global__ __launch_bounds__(256) void consistency_graph_kernel(const double* __restrict__ src,
                                                                const double* __restrict__ dst,
                                                                int n, int words, double beta,
                                                                unsigned long long* __restrict__ bits) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)n * words) return;
  const int i = (int)(gid / words), w = (int)(gid % words);
  const double sx = src[3 * i], sy = src[3 * i + 1], sz = src[3 * i + 2];
  const double dx = dst[3 * i], dy = dst[3 * i + 1], dz = dst[3 * i + 2];
  unsigned long long m = 0;
  for (int b = 0; b < 64; ++b) {
    const int j = w * 64 + b;
    if (j >= n) break;
    if (j == i) continue;
    double e, a = 0, c = 0;
    e = src[3 * j] - sx, a += e * e;
    e = src[3 * j + 1] - sy, a += e * e;
    e = src[3 * j + 2] - sz, a += e * e;
    e = dst[3 * j] - dx, c += e * e;
    e = dst[3 * j + 1] - dy, c += e * e;
    e = dst[3 * j + 2] - dz, c += e * e;
    if (fabs(sqrt(a) - sqrt(c)) <= beta) m |= 1ull << b;
  }
  bits[gid] = m;
}


// ---------------------------------------------------------------------------------------------
// GNC-TLS rotation over the pairwise differences of a clique of m matches (the host loop is
// teaser::gnc_rotation): with m in the thousands the m (m - 1) / 2 differences are millions and
// every one of the up to 100 iterations sums over all of them.  Workgroup (bx, i) owns the
// differences (i, j), j = bx * 256 + lane, j > i, recomputed from the m points each time.  One
// launch does everything an iteration needs from a given rotation: residuals, their maximum, the
// cost under the old weights, the new weights (kept in an m x m array) and the correlation
// matrix under the new weights for the next rotation.  Each workgroup leaves 11 partial values;
// gnc_fold_kernel adds them in a fixed order.
struct GncStep {
  double R[9];
  double th1, th2, c, mu;  // weight = 0 above th1, 1 below th2, sqrt(c / res) - mu between
  int init;                // old weights are 1 (nothing stored yet)
  int update;              // write the new weights and use them for the correlation
};
constexpr int GNC_THREADS = 256;
constexpr int GNC_VALUES = 11;  // H (9), cost, max residual

__global__ __launch_bounds__(GNC_THREADS) void gnc_step_kernel(const double* __restrict__ cs,
                                                               const double* __restrict__ cd,
                                                               int m, GncStep k,
                                                               double* __restrict__ w,
                                                               double* __restrict__ partial) {
  __shared__ double sh[GNC_THREADS / 64][GNC_VALUES];
  const int i = blockIdx.y, j = blockIdx.x * GNC_THREADS + threadIdx.x;
  double v[GNC_VALUES];
#pragma unroll
  for (int q = 0; q < GNC_VALUES; ++q) v[q] = 0.0;
  if (j > i && j < m) {
    double x[3], y[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) x[a] = cs[3 * j + a] - cs[3 * i + a], y[a] = cd[3 * j + a] - cd[3 * i + a];
    double res = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double d = y[a] - (k.R[a * 3] * x[0] + k.R[a * 3 + 1] * x[1] + k.R[a * 3 + 2] * x[2]);
      res += d * d;
    }
    const size_t o = (size_t)i * m + j;
    const double wo = k.init ? 1.0 : w[o];
    double wn = wo;
    if (k.update) {
      wn = res >= k.th1 ? 0.0 : (res <= k.th2 ? 1.0 : sqrt(k.c / res) - k.mu);
      w[o] = wn;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) v[a * 3 + b] = x[a] * wn * y[b];
    v[9] = wo * res;
    v[10] = res;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < GNC_VALUES; ++q) {
    double a = v[q];
    for (int off = 32; off > 0; off >>= 1) {
      const double b = __shfl_xor(a, off);
      a = q == 10 ? fmax(a, b) : a + b;
    }
    if (lane == 0) sh[wave][q] = a;
  }
  __syncthreads();
  if (threadIdx.x < GNC_VALUES) {
    const int q = threadIdx.x;
    double a = sh[0][q];
    for (int wv = 1; wv < GNC_THREADS / 64; ++wv) a = q == 10 ? fmax(a, sh[wv][q]) : a + sh[wv][q];
    partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * GNC_VALUES + q] = a;
  }
}

// out[q] = fold of partial[0..n)[q]: lane t adds rows t, t + 256, ... in order, then the 256
// lane sums are added in lane order
__global__ __launch_bounds__(GNC_THREADS) void gnc_fold_kernel(const double* __restrict__ partial,
                                                               int n, double* __restrict__ out) {
  __shared__ double sh[GNC_THREADS][GNC_VALUES];
  double v[GNC_VALUES];
#pragma unroll
  for (int q = 0; q < GNC_VALUES; ++q) v[q] = 0.0;
  for (int r = threadIdx.x; r < n; r += GNC_THREADS)
#pragma unroll
    for (int q = 0; q < GNC_VALUES; ++q) {
      const double b = partial[(size_t)r * GNC_VALUES + q];
      v[q] = q == 10 ? fmax(v[q], b) : v[q] + b;
    }
#pragma unroll
  for (int q = 0; q < GNC_VALUES; ++q) sh[threadIdx.x][q] = v[q];
  __syncthreads();
  if (threadIdx.x < GNC_VALUES) {
    const int q = threadIdx.x;
    double a = sh[0][q];
    for (int t = 1; t < GNC_THREADS; ++t) a = q == 10 ? fmax(a, sh[t][q]) : a + sh[t][q];
    out[q] = a;
  }
}

}  // namespace lgicp

namespace lgicp_detail {

namespace {
struct Scratch {
  std::vector<void*> ptrs;
  ~Scratch() {
    for (void* p : ptrs) (void)hipFree(p);
  }
  template <typename T>
  T* alloc(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
    ptrs.push_back(p);
    return (T*)p;
  }
};
}  // namespace

// descriptor k-NN table [n_src][k] (device pointers; partial lists live in `d`)
static int launch_feature_knn(lgicp_ctx* ctx, Scratch& d, const float* d_sf, int n_src,
                              const float* d_tf, int n_tgt, int k, int* d_nn, hipStream_t st) {
  const int src_blocks = (n_src + 127) / 128;
  // enough target ranges for ~4 workgroups per CU, each at least four tiles long
  int n_splits = std::max(1, std::min((1024 + src_blocks - 1) / src_blocks,
                                      (n_tgt + 4 * lgicp::FK_TILE - 1) / (4 * lgicp::FK_TILE)));
  int per_split = (n_tgt + n_splits - 1) / n_splits;
  per_split = ((per_split + lgicp::FK_TILE - 1) / lgicp::FK_TILE) * lgicp::FK_TILE;
  n_splits = (n_tgt + per_split - 1) / per_split;
  float* d_pd = d.alloc<float>((size_t)n_splits * n_src * 8);
  int* d_pi = d.alloc<int>((size_t)n_splits * n_src * 8);
  if (!d_pd || !d_pi) return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
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
                     const float guess[16], uint32_t seed, float T_out[16], double* score) {
  std::memcpy(T_out, guess, sizeof(float) * 16);
  *score = DBL_MAX;
  const int n_src = (int)src.idx.size(), n_tgt = (int)tgt.idx.size();
  if (n_src < 3 || n_tgt < 1) return LGICP_OK;
  const int k = std::min(5, n_tgt);  // setCorrespondenceRandomness(5), IcpLoopComputation.cc:697
  hipStream_t st = ctx->stream;
  Scratch d;
  float* d_sf = d.alloc<float>(src.feat.size());
  float* d_tf = d.alloc<float>(tgt.feat.size());
  int* d_nn = d.alloc<int>((size_t)n_src * k);
  float4* d_sk = d.alloc<float4>(n_src);
  float4* d_tk = d.alloc<float4>(n_tgt);
  if (!d_sf || !d_tf || !d_nn || !d_sk || !d_tk) return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  std::vector<float4> hs(n_src), ht(n_tgt);
  for (int i = 0; i < n_src; ++i) hs[i] = make_float4(src.xyz[3 * i], src.xyz[3 * i + 1], src.xyz[3 * i + 2], 1.f);
  for (int i = 0; i < n_tgt; ++i) ht[i] = make_float4(tgt.xyz[3 * i], tgt.xyz[3 * i + 1], tgt.xyz[3 * i + 2], 1.f);
  HIP_TRY(ctx, hipMemcpyAsync(d_sf, src.feat.data(), sizeof(float) * src.feat.size(), hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d_tf, tgt.feat.data(), sizeof(float) * tgt.feat.size(), hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d_sk, hs.data(), sizeof(float4) * n_src, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d_tk, ht.data(), sizeof(float4) * n_tgt, hipMemcpyHostToDevice, st));
  {
    const int rc = launch_feature_knn(ctx, d, d_sf, n_src, d_tf, n_tgt, k, d_nn, st);
    if (rc != LGICP_OK) return rc;
  }
  std::vector<int> nn((size_t)n_src * k);
  HIP_TRY(ctx, hipMemcpyAsync(nn.data(), d_nn, sizeof(int) * nn.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  const sac::Hypotheses hyp = sac::make_hypotheses(src.xyz.data(), n_src, tgt.xyz.data(), nn.data(),
                                                   k, ctx->cfg.sac_iterations, guess, seed);
  const int n_hyp = (int)(hyp.T.size() / 16);
  if (n_hyp == 0) return LGICP_OK;
  float* d_hyp = d.alloc<float>(hyp.T.size());
  const int n_chunks = (n_src + lgicp::SS_QUERIES - 1) / lgicp::SS_QUERIES;
  double* d_ef = d.alloc<double>((size_t)2 * n_hyp * n_chunks);
  if (!d_hyp || !d_ef) return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
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

// descriptor k-NN table [n_src][k] on the device
static int feature_knn_table(lgicp_ctx* ctx, const CloudFeatures& src, const CloudFeatures& tgt,
                             int k, std::vector<int>* nn) {
  const int n_src = (int)src.idx.size(), n_tgt = (int)tgt.idx.size();
  hipStream_t st = ctx->stream;
  Scratch d;
  float* d_sf = d.alloc<float>(src.feat.size());
  float* d_tf = d.alloc<float>(tgt.feat.size());
  int* d_nn = d.alloc<int>((size_t)n_src * k);
  if (!d_sf || !d_tf || !d_nn) return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  HIP_TRY(ctx, hipMemcpyAsync(d_sf, src.feat.data(), sizeof(float) * src.feat.size(), hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d_tf, tgt.feat.data(), sizeof(float) * tgt.feat.size(), hipMemcpyHostToDevice, st));
  {
    const int rc = launch_feature_knn(ctx, d, d_sf, n_src, d_tf, n_tgt, k, d_nn, st);
    if (rc != LGICP_OK) return rc;
  }
  nn->resize((size_t)n_src * k);
  HIP_TRY(ctx, hipMemcpyAsync(nn->data(), d_nn, sizeof(int) * nn->size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return LGICP_OK;
}

// teaser::clique_rotation_host with the sums over the pairwise differences on the device.
// Runs on a stream of the calling thread: several pairs are solved at once from different host
// threads.  Returns false when the device refuses (the caller reports it).
namespace {
// Stream and buffers of the clique rotation search, one set per host thread (several pairs are
// solved at once from different threads) and reused from clique to clique; released when the
// thread ends.  Bound to the device that was current when they were made.
struct CliqueDev {
  int device = -1;
  hipStream_t st = nullptr;
  DevBuf<double> cs, cd, w, partial, out;
  void release() {
    if (device >= 0) {
      int cur = -1;
      (void)hipGetDevice(&cur);
      if (cur != device) (void)hipSetDevice(device);
      cs.release(), cd.release(), w.release(), partial.release(), out.release();
      if (st) (void)hipStreamDestroy(st);
      if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
    }
    st = nullptr;
    device = -1;
  }
  bool bind(int dev) {
    if (device == dev && st) return true;
    release();
    if (hipSetDevice(dev) != hipSuccess) return false;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
      st = nullptr;
      return false;
    }
    device = dev;
    return true;
  }
  ~CliqueDev() { release(); }
};
CliqueDev& thread_clique_dev() {
  thread_local CliqueDev dev;
  return dev;
}
}  // namespace

static bool clique_rotation_device(int device, const std::vector<double>& cs, const std::vector<double>& cd,
                                   const lgicp::teaser::Params& p, double R[9]) {
  namespace tz = lgicp::teaser;
  const int m = (int)(cs.size() / 3);
  for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  CliqueDev& dev = thread_clique_dev();
  if (!dev.bind(device)) return false;
  hipStream_t st = dev.st;
  bool ok = false;
  {
    const dim3 grid((m + lgicp::GNC_THREADS - 1) / lgicp::GNC_THREADS, m);
    const int n_partial = (int)(grid.x * grid.y);
    if (dev.cs.reserve(cs.size()) != hipSuccess || dev.cd.reserve(cd.size()) != hipSuccess ||
        dev.w.reserve((size_t)m * m) != hipSuccess ||
        dev.partial.reserve((size_t)n_partial * lgicp::GNC_VALUES) != hipSuccess ||
        dev.out.reserve(lgicp::GNC_VALUES) != hipSuccess)
      return false;
    double* d_cs = dev.cs.p;
    double* d_cd = dev.cd.p;
    double* d_w = dev.w.p;
    double* d_partial = dev.partial.p;
    double* d_out = dev.out.p;
    double out[lgicp::GNC_VALUES];
    auto step = [&](lgicp::GncStep k) -> bool {
      hipLaunchKernelGGL(lgicp::gnc_step_kernel, grid, dim3(lgicp::GNC_THREADS), 0, st, d_cs, d_cd, m,
                         k, d_w, d_partial);
      hipLaunchKernelGGL(lgicp::gnc_fold_kernel, dim3(1), dim3(lgicp::GNC_THREADS), 0, st, d_partial,
                         n_partial, d_out);
      return hipGetLastError() == hipSuccess &&
             hipMemcpyAsync(out, d_out, sizeof(out), hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipStreamSynchronize(st) == hipSuccess;
    };
    if (hipMemcpyAsync(d_cs, cs.data(), sizeof(double) * cs.size(), hipMemcpyHostToDevice, st) == hipSuccess &&
        hipMemcpyAsync(d_cd, cd.data(), sizeof(double) * cd.size(), hipMemcpyHostToDevice, st) == hipSuccess) {
      ok = true;
      double nb2 = p.noise_bound * p.noise_bound;
      if (nb2 < 1e-16) nb2 = 1e-2;
      lgicp::GncStep k{};
      for (int i = 0; i < 9; ++i) k.R[i] = R[i];
      k.init = 1;
      // iteration 0 needs the largest residual under the unweighted rotation before it can
      // set its weights: correlation of unit weights, then the residuals, then the update
      ok = ok && step(k);
      if (ok) {
        tz::rotation_from_correlation(out, R);
        for (int i = 0; i < 9; ++i) k.R[i] = R[i];
        ok = step(k);
      }
      double mu = ok ? 1.0 / (2.0 * out[10] / nb2 - 1.0) : 0.0;
      double prev_cost = INFINITY;
      for (int it = 0; ok && mu > 0 && it < p.rotation_max_iterations; ++it) {
        if (it > 0) {
          tz::rotation_from_correlation(out, R);
          for (int i = 0; i < 9; ++i) k.R[i] = R[i];
          k.init = 0;
        }
        k.update = 1;
        k.th1 = (mu + 1) / mu * nb2;
        k.th2 = mu / (mu + 1) * nb2;
        k.c = nb2 * mu * (mu + 1);
        k.mu = mu;
        ok = step(k);
        const double cost = out[9];
        const double cost_diff = std::fabs(cost - prev_cost);
        mu *= p.rotation_gnc_factor;
        prev_cost = cost;
        if (cost_diff < p.rotation_cost_threshold) break;
      }
    }
  }
  // a failed call may have left work queued on the stream: nothing may still read the buffers
  // when the next clique overwrites them
  if (!ok) (void)hipStreamSynchronize(st);
  return ok;
}

// GetTeaserInitialAlignment from the keypoints on (IcpLoopComputation.cc:990-1081), first half:
// descriptor matches and their scale-consistency graph, both built on the device.
int teaser_prepare_device(lgicp_ctx* ctx, const CloudFeatures& src, const CloudFeatures& tgt,
                          TeaserJob* job) {
  namespace tz = lgicp::teaser;
  job->n_matches = 0;
  job->src.clear();
  job->dst.clear();
  job->graph = tz::BitGraph();
  const int n_src = (int)src.idx.size(), n_tgt = (int)tgt.idx.size();
  if (n_src == 0 || n_tgt == 0) return LGICP_OK;
  const int k = std::min(5, n_tgt);  // calculateKCorrespondences(..., 5), IcpLoopComputation.cc:997
  std::vector<int> nn;
  int rc = feature_knn_table(ctx, src, tgt, k, &nn);
  if (rc != LGICP_OK) return rc;
  const int n = n_src * k;
  job->n_matches = n;
  if (n <= 10) return LGICP_OK;  // IcpLoopComputation.cc:1002
  // the consistency graph is a dense n x n bit matrix: 65536 matches are 512 MiB of rows
  constexpr int kMaxMatches = 65536;
  if (n > kMaxMatches) {
    // a per-pair outcome, not a call-level error: the pair keeps its odometry guess (what the
    // reference does whenever TEASER++ does not produce a better transform,
    // IcpLoopComputation.cc:1065-1074) and the other pairs of the submit go on
    if (std::getenv("LGICP_TRACE"))
      std::fprintf(stderr, "[lgicp] teaser: %d descriptor matches exceed the %d the consistency graph "
                   "is sized for; the pair keeps its guess\n", n, kMaxMatches);
    return LGICP_OK;
  }
  std::vector<double>& s = job->src;
  std::vector<double>& d = job->dst;
  s.resize((size_t)3 * n);
  d.resize((size_t)3 * n);
  for (int i = 0; i < n_src; ++i)
    for (int q = 0; q < k; ++q)
      for (int c = 0; c < 3; ++c) {
        s[3 * ((size_t)i * k + q) + c] = (double)src.xyz[3 * i + c];
        d[3 * ((size_t)i * k + q) + c] = (double)tgt.xyz[3 * nn[(size_t)i * k + q] + c];
      }
  const tz::Params prm;
  tz::BitGraph& g = job->graph;
  g.n = n;
  g.words = (n + 63) / 64;
  const size_t n_words = (size_t)n * g.words;
  hipStream_t st = ctx->stream;
  Scratch dev;
  double* d_s = dev.alloc<double>(s.size());
  double* d_d = dev.alloc<double>(d.size());
  if (!d_s || !d_d || job->d_bits.reserve(n_words) != hipSuccess)
    return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  unsigned long long* d_bits = job->d_bits.p;
  HIP_TRY(ctx, hipMemcpyAsync(d_s, s.data(), sizeof(double) * s.size(), hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d_d, d.data(), sizeof(double) * d.size(), hipMemcpyHostToDevice, st));
  const long long total = (long long)n * g.words;
  hipLaunchKernelGGL(lgicp::consistency_graph_kernel, dim3((unsigned)((total + 255) / 256)),
                     dim3(256), 0, st, d_s, d_d, n, g.words,
                     2.0 * prm.noise_bound * std::sqrt(prm.cbar2), d_bits);
  HIP_TRY(ctx, hipGetLastError());
  // The graph stays on the device: core numbers and the clique search run there
  // (teaser_graph.hip).  Only a graph too large for that path (or reserved[4] = -1: the host
  // search, for cross-checks) is brought to the host.
  g.bits.clear();
  if (n > clique_device_max_vertices() || ctx->cfg.reserved[4] == -1) {
    g.bits.resize(n_words);
    HIP_TRY(ctx, hipMemcpyAsync(g.bits.data(), d_bits, sizeof(uint64_t) * n_words, hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return LGICP_OK;
}

// Second half (safe to run for several pairs at once from different threads): clique, rotation,
// translation and the acceptance test.  Graph work and 3x3 algebra stay on the host; the sums
// over the clique's pairwise differences go to the device once the clique is large enough to
// pay for the launches.  T_out is the guess unless the TEASER++ transform scores at least as
// well over the matches.
int teaser_finish(lgicp_ctx* ctx, const TeaserJob& job, const float guess[16], float T_out[16],
                  int* used, int* n_clique) {
  namespace tz = lgicp::teaser;
  std::memcpy(T_out, guess, sizeof(float) * 16);
  *used = *n_clique = 0;
  if (job.graph.n == 0) return LGICP_OK;
  const tz::Params prm;
  bool device_ok = true;
  std::vector<int> clique;
  if (job.graph.bits.empty()) {
    CliqueDev& cd = thread_clique_dev();
    if (!cd.bind(ctx->cfg.device) ||
        !greedy_clique_device(ctx->cfg.device, cd.st, job.d_bits.p, job.graph.n, job.graph.words, &clique))
      return fail(ctx, LGICP_E_HIP, "TEASER++ initialisation: clique search over %d matches failed on the "
                  "device (%s)", job.graph.n, hipGetErrorString(hipGetLastError()));
  } else {
    // large graphs: a pair with ten thousand matches would otherwise keep one thread busy for
    // hundreds of milliseconds after every other pair of the submit has finished
    clique = tz::greedy_clique(job.graph, job.graph.n >= 2048 ? 8 : 1);
  }
  const tz::Solution sol = tz::solve_on_clique(
      job.src, job.dst, std::move(clique), prm,
      [&](const std::vector<double>& cs, const std::vector<double>& cd, const tz::Params& p, double R[9]) {
        if (cs.size() / 3 < 128)
          tz::clique_rotation_host(cs, cd, p, R);
        else
          device_ok = clique_rotation_device(ctx->cfg.device, cs, cd, p, R);
      });
  if (!device_ok)
    return fail(ctx, LGICP_E_HIP, "TEASER++ initialisation: rotation search over a clique of %d matches "
                "failed on the device (%s)", (int)sol.clique.size(), hipGetErrorString(hipGetLastError()));
  *n_clique = (int)sol.clique.size();
  double Ro[9], to[3];
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) Ro[a * 3 + b] = (double)guess[a * 4 + b];
    to[a] = (double)guess[a * 4 + 3];
  }
  const double e_teaser = tz::error_metric(job.src, job.dst, sol.R, sol.t, 100.0);
  const double e_odom = tz::error_metric(job.src, job.dst, Ro, to, 100.0);
  if (e_teaser <= e_odom) {  // IcpLoopComputation.cc:1065-1068
    for (int i = 0; i < 16; ++i) T_out[i] = (i == 15) ? 1.f : 0.f;
    for (int a = 0; a < 3; ++a) {
      for (int b = 0; b < 3; ++b) T_out[a * 4 + b] = (float)sol.R[a * 3 + b];
      T_out[a * 4 + 3] = (float)sol.t[a];
    }
    *used = 1;
  }
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
  return teaser_finish(ctx, job, guess, T_out, used, n_clique);
}

}  // namespace lgicp_detail
