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
// the cloud and an exhaustive tiled sweep beats any tree.  One lane per candidate, not per point: a
// wave owns four points at a time, its 64 lanes test 64 candidates of the LDS tile against each of
// them, and the few within the radius go into the point's queue; whenever a queue holds 64 the
// lanes take one neighbour each and compute its pair features together (with one lane per point the
// ~200 instructions of a pair ran for the two or three lanes of a wave that had found a neighbour
// at that candidate).  The bins are counters in LDS, so the order in which neighbours arrive does
// not matter, and a 16k-point cloud is 1,024 workgroups.
#include <algorithm>

#include "lgicp_device.h"

namespace lgicp {

static constexpr int FP_TILE = 512;
static constexpr int FP_BINS = 11;

// need[j] = 1 for every surface point within the radius of some keypoint.  need starts at zero (the
// caller's memset); workgroup (x, y) looks at the keypoints of range y only and writes ones, so the
// ranges meet in any order.
__global__ void fpfh_need_kernel(const float4* __restrict__ surf, int n,
                                 const float4* __restrict__ keys, int n_key, int per_range, float r2,
                                 unsigned char* __restrict__ need) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const float4 p = surf[j];
  bool f = false;
  const int k_end = min(n_key, (int)(blockIdx.y + 1) * per_range);
  for (int k = blockIdx.y * per_range; k < k_end; ++k) {
    const float4 q = keys[k];  // wave-uniform
    const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
    const float d = (dx * dx + dy * dy) + dz * dz;
    if (d < r2) f = true;
  }
  if (f) need[j] = 1;
}

__device__ __forceinline__ int clamp_bin(float x) {
  int h = (int)floorf(x);
  h = h < 0 ? 0 : h;
  return h >= FP_BINS ? FP_BINS - 1 : h;
}

static constexpr int SP_PW = 4;       // points a wave sweeps for at a time
static constexpr int SP_QUEUE = 128;  // neighbours waiting per point (64 trigger a round, 64 can arrive at once)

// spfh[i][33] of every surface point with need[i] set; the other rows are left alone.
__global__ __launch_bounds__(256) void spfh_kernel(const float4* __restrict__ surf,
                                                   const float4* __restrict__ nrm, int n,
                                                   const unsigned char* __restrict__ need,
                                                   float r2, const float4* __restrict__ boxes,
                                                   float* __restrict__ spfh) {
  __shared__ __attribute__((aligned(16))) float tile[3 * FP_TILE];
  __shared__ int hist[4][SP_PW][33];
  __shared__ int queue[4][SP_PW][SP_QUEUE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = (blockIdx.x * 4 + wave) * SP_PW;  // this wave's points: i0 .. i0 + SP_PW - 1
  bool live[SP_PW];
  float4 p[SP_PW], a[SP_PW];
  int waiting[SP_PW], n_nbr[SP_PW];
  bool any = false;
#pragma unroll
  for (int q = 0; q < SP_PW; ++q) {
    live[q] = i0 + q < n && need[i0 + q];
    p[q] = a[q] = make_float4(0, 0, 0, 0);
    if (live[q]) p[q] = surf[i0 + q], a[q] = nrm[i0 + q];
    waiting[q] = n_nbr[q] = 0;
    any = any || live[q];
  }
  // skip blocks in which no point needs an SPFH
  if (__syncthreads_count(any) == 0) return;
  for (int e = lane; e < SP_PW * 33; e += 64) hist[wave][0][e] = 0;
  // the first `count` neighbours waiting for point q, one per lane: pcl::computePairFeatures(p1 =
  // the point, p2 = the neighbour) and the three bins
  auto pair_features = [&](int q, int count) {
    if (lane >= count) return;
    const int j = queue[wave][q][lane];
    if (j == i0 + q) return;
    const float4 pj = surf[j], mj = nrm[j];
    float dx = pj.x - p[q].x, dy = pj.y - p[q].y, dz = pj.z - p[q].z;
    const float f4 = sqrtf(dx * dx + dy * dy + dz * dz);
    if (f4 == 0.0f) return;
    float n1x = a[q].x, n1y = a[q].y, n1z = a[q].z;
    float n2x = mj.x, n2y = mj.y, n2z = mj.z;
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
    atomicAdd(&hist[wave][q][b1], 1);
    atomicAdd(&hist[wave][q][FP_BINS + b2], 1);
    atomicAdd(&hist[wave][q][2 * FP_BINS + b3], 1);
  };
  for (int base = 0; base < n; base += FP_TILE) {
    // chunks of 64 consecutive surface points whose box lies out of reach are not looked at
    // (harris.hip); the bins are counters, the order of arrival does not matter anyway
    unsigned reach[SP_PW], reach_any = 0;
#pragma unroll
    for (int q = 0; q < SP_PW; ++q) {
      reach[q] = live[q] ? chunks_in_reach(boxes, n, base, FP_TILE / 64, lane, p[q].x, p[q].y, p[q].z, r2) : 0u;
      reach_any |= reach[q];
    }
    if (!__syncthreads_or(reach_any != 0)) continue;  // (also the barrier between two tiles)
    for (int j = tid; j < FP_TILE; j += 256) {
      float4 c = make_float4(PAD_COORD, PAD_COORD, PAD_COORD, 0);
      if (base + j < n) c = surf[base + j];
      tile[j] = c.x;
      tile[FP_TILE + j] = c.y;
      tile[2 * FP_TILE + j] = c.z;
    }
    __syncthreads();
    const int jend = min(FP_TILE, n - base);  // (the rest of the tile is padding, never within the radius)
    for (int c = 0; c < jend; c += 64) {
      const unsigned bit = 1u << (c >> 6);
      if (!(reach_any & bit)) continue;  // wave-uniform
      const float tx = tile[c + lane], ty = tile[FP_TILE + c + lane], tz = tile[2 * FP_TILE + c + lane];
#pragma unroll
      for (int q = 0; q < SP_PW; ++q) {
        if (!(reach[q] & bit)) continue;  // wave-uniform (never set for a point that is not live)
        // radius test on the search distance (query - point, x, y, z order)
        const float ex = p[q].x - tx, ey = p[q].y - ty, ez = p[q].z - tz;
        const bool in = (ex * ex + ey * ey) + ez * ez < r2;
        const unsigned long long m = __ballot(in);
        if (m == 0) continue;
        if (in) queue[wave][q][waiting[q] + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = base + c + lane;
        const int came = __builtin_popcountll(m);
        waiting[q] += came, n_nbr[q] += came;
        if (waiting[q] >= 64) {
          pair_features(q, 64);
          const int rest = waiting[q] - 64;  // < 64: moves to the front
          int moved = 0;
          if (lane < rest) moved = queue[wave][q][64 + lane];
          if (lane < rest) queue[wave][q][lane] = moved;
          waiting[q] = rest;
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < SP_PW; ++q) {
    if (!live[q]) continue;
    pair_features(q, waiting[q]);
    // every admitted pair adds hist_incr = 100 / (k - 1); the reference adds them one by one in
    // float, here count * incr (differs by float accumulation error only)
    const float incr = 100.0f / (float)(n_nbr[q] - 1);
    if (lane < 33) spfh[(size_t)(i0 + q) * 33 + lane] = (float)hist[wave][q][lane] * incr;
  }
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
                       float r2, unsigned char* need, float* spfh, float* out, const float4* boxes,
                       hipStream_t stream) {
  if (n_key <= 0) return hipSuccess;
  if (n > 0) {
    // (rows of points that need no SPFH stay zero)
    hipError_t e = hipMemsetAsync(spfh, 0, sizeof(float) * 33 * (size_t)n, stream);
    if (e == hipSuccess) e = hipMemsetAsync(need, 0, (size_t)n, stream);
    if (e != hipSuccess) return e;
    // ranges of the keypoints so that the launch fills the GPU (about 2,000 workgroups)
    const int blocks = (n + 255) / 256;
    int ranges = std::max(1, std::min((2048 + blocks - 1) / blocks, (n_key + 255) / 256));
    const int per_range = (n_key + ranges - 1) / ranges;
    ranges = (n_key + per_range - 1) / per_range;
    hipLaunchKernelGGL(fpfh_need_kernel, dim3(blocks, ranges), dim3(256), 0, stream, surf, n, keys, n_key,
                       per_range, r2, need);
    hipLaunchKernelGGL(spfh_kernel, dim3((n + 4 * SP_PW - 1) / (4 * SP_PW)), dim3(256), 0, stream, surf, nrm, n,
                       need, r2, boxes, spfh);
  }
  hipLaunchKernelGGL(fpfh_weight_kernel, dim3(n_key), dim3(256), 0, stream, surf, n, keys, r2,
                     spfh, out);
  return hipGetLastError();
}

}  // namespace lgicp
