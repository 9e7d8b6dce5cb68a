// fpfh.hip — FPFH descriptors (33 bins) at keypoints over a full search surface, for the
// SAC-IA / TEASER++ global-registration front end.
//
// Replaces lamp_utils::ComputeFeatures (lamp_utils/src/PointCloudUtils.cc:98-113), i.e. PCL's
// FPFHEstimationOMP with input = keypoints, search surface = the whole cloud and a radius
// search (sac_features_radius = 2.0 m): (1) every surface point that lies within the radius of
// some keypoint gets an SPFH — the histogram of the Darboux-frame angles (f1, f2, f3) between
// it and each of its own radius neighbours, 11 bins each, increment 100/(k-1); (2) the FPFH of
// a keypoint is the 1/d^2-weighted sum of its neighbours' SPFHs, each 11-bin block rescaled
// to sum 100.
//
// Mapping: the radius is large against the scene, so a neighbourhood holds a sizeable share of
// the cloud and an exhaustive tiled sweep (same LDS staging as the 1-NN kernel, points plus
// normals) beats any tree: one lane per surface point, per-lane bin counters in LDS laid out
// [bin][lane] so that lanes never share a bank.  One lane per point alone would be 64 workgroups
// for a 16k-point cloud, a wave on every fourth SIMD; the bins are counters, so the sweep of a
// point is cut into ranges of the surface (gridDim.y) whose counts meet in integer atomics, and
// the float histogram is made from the totals afterwards: the same numbers in any order.
#include <algorithm>

#include "lgicp_device.h"

namespace lgicp {

static constexpr int FP_TILE = 512;
static constexpr int FP_BINS = 11;

__global__ void fpfh_need_kernel(const float4* __restrict__ surf, int n,
                                 const float4* __restrict__ keys, int n_key, float r2,
                                 unsigned char* __restrict__ need) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const float4 p = surf[j];
  unsigned char f = 0;
  for (int k = 0; k < n_key; ++k) {
    const float4 q = keys[k];  // wave-uniform
    const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
    const float d = (dx * dx + dy * dy) + dz * dz;
    if (d < r2) f = 1;
  }
  need[j] = f;
}

__device__ __forceinline__ int clamp_bin(float x) {
  int h = (int)floorf(x);
  h = h < 0 ? 0 : h;
  return h >= FP_BINS ? FP_BINS - 1 : h;
}

// counts[i][33] (zeroed by the caller, the float histogram's memory) and nbr[i] receive the range's
// share of point i's bins and of its neighbour count.  Range blockIdx.y = surface points
// [blockIdx.y * per_range, (blockIdx.y + 1) * per_range), per_range a multiple of FP_TILE.
__global__ __launch_bounds__(256) void spfh_kernel(const float4* __restrict__ surf,
                                                   const float4* __restrict__ nrm, int n,
                                                   const unsigned char* __restrict__ need,
                                                   float r2, int per_range, int* __restrict__ counts,
                                                   int* __restrict__ nbr) {
  __shared__ __attribute__((aligned(16))) float tile[6 * FP_TILE];
  __shared__ int cnt[33 * 256];
  const int tid = threadIdx.x;
  const int i = blockIdx.x * 256 + tid;
  const bool live = i < n && need[i];
  // skip blocks in which no point needs an SPFH
  if (__syncthreads_count(live) == 0) return;
  float4 p = make_float4(0, 0, 0, 0), a = p;
  if (i < n) {
    p = surf[i];
    a = nrm[i];
  }
#pragma unroll
  for (int b = 0; b < 33; ++b) cnt[b * 256 + tid] = 0;
  int n_nbr = 0;
  const int range_end = min(n, (int)(blockIdx.y + 1) * per_range);
  for (int base = blockIdx.y * per_range; base < range_end; base += FP_TILE) {
    __syncthreads();
    for (int j = tid; j < FP_TILE; j += 256) {
      float4 q = make_float4(PAD_COORD, PAD_COORD, PAD_COORD, 0), m = make_float4(0, 0, 0, 0);
      if (base + j < range_end) {
        q = surf[base + j];
        m = nrm[base + j];
      }
      tile[j] = q.x;
      tile[FP_TILE + j] = q.y;
      tile[2 * FP_TILE + j] = q.z;
      tile[3 * FP_TILE + j] = m.x;
      tile[4 * FP_TILE + j] = m.y;
      tile[5 * FP_TILE + j] = m.z;
    }
    __syncthreads();
    if (!live) continue;
    // Four candidates per step (one 16-byte LDS read per coordinate, four independent chains of the
    // radius test on the search distance: query - point, x, y, z order) and a branch only when one of
    // them is a neighbour; the tail of the last tile holds padding points, never within the radius.
    auto neighbour = [&](int j) {
      float dx = tile[j] - p.x, dy = tile[FP_TILE + j] - p.y, dz = tile[2 * FP_TILE + j] - p.z;
      ++n_nbr;
      if (base + j == i) return;
      // pcl::computePairFeatures(p1 = this point, p2 = neighbour)
      const float f4 = sqrtf(dx * dx + dy * dy + dz * dz);
      if (f4 == 0.0f) return;
      float n1x = a.x, n1y = a.y, n1z = a.z;
      float n2x = tile[3 * FP_TILE + j], n2y = tile[4 * FP_TILE + j], n2z = tile[5 * FP_TILE + j];
      const float angle1 = (n1x * dx + n1y * dy + n1z * dz) / f4;
      const float angle2 = (n2x * dx + n2y * dy + n2z * dz) / f4;
      float f3;
      if (acosf(fabsf(angle1)) > acosf(fabsf(angle2))) {
        float t;
        t = n1x, n1x = n2x, n2x = t;
        t = n1y, n1y = n2y, n2y = t;
        t = n1z, n1z = n2z, n2z = t;
        dx *= -1.f, dy *= -1.f, dz *= -1.f;
        f3 = -angle2;
      } else {
        f3 = angle1;
      }
      float vx = dy * n1z - dz * n1y, vy = dz * n1x - dx * n1z, vz = dx * n1y - dy * n1x;
      const float v_norm = sqrtf(vx * vx + vy * vy + vz * vz);
      if (v_norm == 0.0f) return;
      const float s = 1.0f / v_norm;
      vx *= s, vy *= s, vz *= s;
      const float wx = n1y * vz - n1z * vy, wy = n1z * vx - n1x * vz, wz = n1x * vy - n1y * vx;
      const float f2 = vx * n2x + vy * n2y + vz * n2z;
      const float f1 = atan2f(wx * n2x + wy * n2y + wz * n2z, n1x * n2x + n1y * n2y + n1z * n2z);
      const int b1 = clamp_bin((float)(FP_BINS * ((f1 + M_PI) * (1.0 / (2.0 * M_PI)))));
      const int b2 = clamp_bin((float)(FP_BINS * ((f2 + 1.0) * 0.5)));
      const int b3 = clamp_bin((float)(FP_BINS * ((f3 + 1.0) * 0.5)));
      cnt[b1 * 256 + tid] += 1;
      cnt[(FP_BINS + b2) * 256 + tid] += 1;
      cnt[(2 * FP_BINS + b3) * 256 + tid] += 1;
    };
    const int jend = (min(FP_TILE, range_end - base) + 3) & ~3;
    for (int j = 0; j < jend; j += 4) {
      const float4 X = *reinterpret_cast<const float4*>(tile + j);
      const float4 Y = *reinterpret_cast<const float4*>(tile + FP_TILE + j);
      const float4 Z = *reinterpret_cast<const float4*>(tile + 2 * FP_TILE + j);
      const float tx[4] = {X.x, X.y, X.z, X.w}, ty[4] = {Y.x, Y.y, Y.z, Y.w}, tz[4] = {Z.x, Z.y, Z.z, Z.w};
      bool in[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ex = p.x - tx[e], ey = p.y - ty[e], ez = p.z - tz[e];
        in[e] = (ex * ex + ey * ey) + ez * ez < r2;
      }
      if (!(in[0] || in[1] || in[2] || in[3])) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (in[e]) neighbour(j + e);
    }
  }
  if (!live) return;
  int* o = counts + (size_t)i * 33;
#pragma unroll
  for (int b = 0; b < 33; ++b) {
    const int c = cnt[b * 256 + tid];
    if (c) atomicAdd(o + b, c);
  }
  if (n_nbr) atomicAdd(nbr + i, n_nbr);
}

// every admitted pair adds hist_incr = 100 / (k - 1); the reference adds them one by one in
// float, here count * incr (differs by float accumulation error only).  In place: the counts of
// a point become its histogram.
__global__ void spfh_finish_kernel(int n, const unsigned char* __restrict__ need, const int* __restrict__ nbr,
                                   float* __restrict__ spfh) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * 33) return;
  const int i = e / 33;
  if (!need[i]) return;
  const float incr = 100.0f / (float)(nbr[i] - 1);
  spfh[e] = (float)reinterpret_cast<const int*>(spfh)[e] * incr;
}

__global__ __launch_bounds__(256) void fpfh_weight_kernel(const float4* __restrict__ surf, int n,
                                                          const float4* __restrict__ keys,
                                                          float r2,
                                                          const float* __restrict__ spfh,
                                                          float* __restrict__ out) {
  __shared__ double sh[4][33];
  const float4 q = keys[blockIdx.x];
  double acc[33];
#pragma unroll
  for (int b = 0; b < 33; ++b) acc[b] = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) {
    const float4 p = surf[j];
    const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    if (!(d2 < r2) || d2 == 0.f) continue;  // "minus the query point itself"
    const float w = 1.0f / d2;
    const float* h = spfh + (size_t)j * 33;
#pragma unroll
    for (int b = 0; b < 33; ++b) acc[b] += (double)(h[b] * w);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int b = 0; b < 33; ++b) {
    double v = acc[b];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) sh[wave][b] = v;
  }
  __syncthreads();
  if (threadIdx.x < 33) {
    const double v = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
    sh[0][threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x < 33) {
    const int g = threadIdx.x / FP_BINS;
    double sum = 0;
    for (int b = 0; b < FP_BINS; ++b) sum += sh[0][g * FP_BINS + b];
    if (sum != 0) sum = 100.0 / sum;
    out[(size_t)blockIdx.x * 33 + threadIdx.x] = (float)sh[0][threadIdx.x] * (float)sum;
  }
}

hipError_t launch_fpfh(const float4* surf, const float4* nrm, int n, const float4* keys, int n_key,
                       float r2, unsigned char* need, float* spfh, int* nbr, float* out,
                       hipStream_t stream) {
  if (n_key <= 0) return hipSuccess;
  if (n > 0) {
    hipLaunchKernelGGL(fpfh_need_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, surf, n, keys,
                       n_key, r2, need);
    // ranges of the surface so that the sweep fills the GPU (about 2,000 workgroups), each at least
    // four tiles long
    const int blocks = (n + 255) / 256;
    int ranges = std::max(1, std::min((2048 + blocks - 1) / blocks, (n + 4 * FP_TILE - 1) / (4 * FP_TILE)));
    int per_range = (n + ranges - 1) / ranges;
    per_range = ((per_range + FP_TILE - 1) / FP_TILE) * FP_TILE;
    ranges = (n + per_range - 1) / per_range;
    hipError_t e = hipMemsetAsync(spfh, 0, sizeof(float) * 33 * (size_t)n, stream);
    if (e == hipSuccess) e = hipMemsetAsync(nbr, 0, sizeof(int) * (size_t)n, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(spfh_kernel, dim3(blocks, ranges), dim3(256), 0, stream, surf, nrm, n, need, r2,
                       per_range, reinterpret_cast<int*>(spfh), nbr);
    hipLaunchKernelGGL(spfh_finish_kernel, dim3((n * 33 + 255) / 256), dim3(256), 0, stream, n, need, nbr, spfh);
  }
  hipLaunchKernelGGL(fpfh_weight_kernel, dim3(n_key), dim3(256), 0, stream, surf, n, keys, r2,
                     spfh, out);
  return hipGetLastError();
}

}  // namespace lgicp
