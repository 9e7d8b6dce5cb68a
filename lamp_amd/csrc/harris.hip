// harris.hip — Harris3D keypoints on the normals of a cloud, for the SAC-IA / TEASER++ front end.
//
// Replaces lamp_utils::ComputeKeypoints with precomputed normals
// (lamp_utils/src/PointCloudUtils.cc:78-96): pcl::HarrisKeypoint3D, HARRIS response, radius
// search, threshold, non-maxima suppression, no refinement.  Response of a point = 0.04 + det(C)
// - 0.04 trace(C)^2 in float, C = mean of n n^T over the neighbours within the radius (finite
// normals only); a point is a keypoint when its response is finite, >= threshold and no
// neighbour within the radius has a strictly larger response.
//
// Mapping: two exhaustive tiled sweeps (response, then suppression), one lane per point, the
// cloud streaming through LDS; the radius test uses the search distance arithmetic of the 1-NN
// kernel and FLANN's strict `<`.  Neighbours are summed in ascending index (PCL visits them by
// ascending distance), so responses can differ from PCL's in the last float bits.
#include <algorithm>

#include "lgicp_device.h"

namespace lgicp {

static constexpr int HR_TILE = 512;

__global__ __launch_bounds__(256) void harris_response_kernel(const float4* __restrict__ xyz,
                                                              const float4* __restrict__ nrm,
                                                              int n, float r2,
                                                              float* __restrict__ response) {
  __shared__ __attribute__((aligned(16))) float tile[6 * HR_TILE];
  const int tid = threadIdx.x;
  const int i = blockIdx.x * 256 + tid;
  float4 p = make_float4(__builtin_nanf(""), 0, 0, 0);
  if (i < n) p = xyz[i];
  float xx = 0, xy = 0, xz = 0, yx = 0, yy = 0, yz = 0, zz = 0;
  unsigned count = 0;
  for (int base = 0; base < n; base += HR_TILE) {
    __syncthreads();
    for (int j = tid; j < HR_TILE; j += 256) {
      float4 q = make_float4(PAD_COORD, PAD_COORD, PAD_COORD, 0), m = make_float4(0, 0, 0, 0);
      if (base + j < n) {
        q = xyz[base + j];
        m = nrm[base + j];
      }
      tile[j] = q.x;
      tile[HR_TILE + j] = q.y;
      tile[2 * HR_TILE + j] = q.z;
      tile[3 * HR_TILE + j] = m.x;
      tile[4 * HR_TILE + j] = m.y;
      tile[5 * HR_TILE + j] = m.z;
    }
    __syncthreads();
    // Four candidates per step (one 16-byte LDS read per coordinate, four independent distance
    // chains) and a branch only when one of them lies within the radius, which few do; the
    // admitted ones are added in ascending index as before.  The tail of the last tile holds
    // padding points, which are never within the radius.
    const int jend = (min(HR_TILE, n - base) + 3) & ~3;
    for (int j = 0; j < jend; j += 4) {
      const float4 X = *reinterpret_cast<const float4*>(tile + j);
      const float4 Y = *reinterpret_cast<const float4*>(tile + HR_TILE + j);
      const float4 Z = *reinterpret_cast<const float4*>(tile + 2 * HR_TILE + j);
      const float tx[4] = {X.x, X.y, X.z, X.w}, ty[4] = {Y.x, Y.y, Y.z, Y.w}, tz[4] = {Z.x, Z.y, Z.z, Z.w};
      bool in[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dx = p.x - tx[e], dy = p.y - ty[e], dz = p.z - tz[e];
        in[e] = (dx * dx + dy * dy) + dz * dz < r2;
      }
      if (!(in[0] || in[1] || in[2] || in[3])) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (!in[e]) continue;
        const float mx = tile[3 * HR_TILE + j + e], my = tile[4 * HR_TILE + j + e], mz = tile[5 * HR_TILE + j + e];
        if (!isfinite(mx)) continue;
        xx += mx * mx;
        xy += my * mx;
        xz += mz * mx;
        yx += mx * my;
        yy += my * my;
        yz += mz * my;
        zz += mz * mz;
        ++count;
      }
    }
  }
  if (i >= n) return;
  float r = 0.f;
  if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
    if (count > 0) {
      const float c = (float)count;
      xx = xx / c, xy = xy / c, xz = xz / c, yx = yx / c, yy = yy / c, yz = yz / c, zz = zz / c;
    }
    (void)yx;
    const float trace = xx + yy + zz;
    if (trace != 0) {
      const float det = xx * yy * zz + 2.0f * xy * xz * yz - xz * xz * yy - xy * xy * zz - yz * yz * xx;
      r = 0.04f + det - 0.04f * trace * trace;
    }
  }
  response[i] = r;
}

// The suppression sweep of a point is cut into ranges of the cloud (gridDim.y; one lane per point
// alone leaves three SIMDs in four idle on a 16k-point cloud): is_key starts at 1 everywhere (the
// caller's memset) and every range clears the points it rules out, so all writes are zeros.
// Range blockIdx.y = points [blockIdx.y * per_range, (blockIdx.y + 1) * per_range), per_range a
// multiple of HR_TILE.
__global__ __launch_bounds__(256) void harris_nms_kernel(const float4* __restrict__ xyz,
                                                         const float* __restrict__ response,
                                                         int n, float r2, float threshold, int per_range,
                                                         unsigned char* __restrict__ is_key) {
  __shared__ __attribute__((aligned(16))) float tile[4 * HR_TILE];
  const int tid = threadIdx.x;
  const int i = blockIdx.x * 256 + tid;
  float4 p = make_float4(__builtin_nanf(""), 0, 0, 0);
  float ri = -1.f;
  if (i < n) {
    p = xyz[i];
    ri = response[i];
  }
  bool cand = i < n && isfinite(p.x) && isfinite(p.y) && isfinite(p.z) && isfinite(ri) &&
              !(ri < threshold);
  if (i < n && !cand) is_key[i] = 0;
  if (__syncthreads_count(cand) == 0) return;
  const int range_end = min(n, (int)(blockIdx.y + 1) * per_range);
  for (int base = blockIdx.y * per_range; base < range_end; base += HR_TILE) {
    __syncthreads();
    for (int j = tid; j < HR_TILE; j += 256) {
      float4 q = make_float4(PAD_COORD, PAD_COORD, PAD_COORD, 0);
      float rj = -1.f;
      if (base + j < range_end) {
        q = xyz[base + j];
        rj = response[base + j];
      }
      tile[j] = q.x;
      tile[HR_TILE + j] = q.y;
      tile[2 * HR_TILE + j] = q.z;
      tile[3 * HR_TILE + j] = rj;
    }
    __syncthreads();
    if (!cand) continue;
    const int jend = min(HR_TILE, range_end - base);
    for (int j = 0; j < jend; ++j) {
      const float dx = p.x - tile[j], dy = p.y - tile[HR_TILE + j], dz = p.z - tile[2 * HR_TILE + j];
      const float d2 = (dx * dx + dy * dy) + dz * dz;
      if (d2 < r2 && ri < tile[3 * HR_TILE + j]) cand = false;
    }
  }
  if (i < n && !cand) is_key[i] = 0;
}

hipError_t launch_harris(const float4* xyz, const float4* nrm, int n, float r2, float threshold,
                         float* response, unsigned char* is_key, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const dim3 grid((n + 255) / 256), block(256);
  hipLaunchKernelGGL(harris_response_kernel, grid, block, 0, stream, xyz, nrm, n, r2, response);
  const int blocks = (n + 255) / 256;
  int ranges = std::max(1, std::min((2048 + blocks - 1) / blocks, (n + 4 * HR_TILE - 1) / (4 * HR_TILE)));
  int per_range = (n + ranges - 1) / ranges;
  per_range = ((per_range + HR_TILE - 1) / HR_TILE) * HR_TILE;
  ranges = (n + per_range - 1) / per_range;
  const hipError_t e = hipMemsetAsync(is_key, 1, (size_t)n, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(harris_nms_kernel, dim3(blocks, ranges), block, 0, stream, xyz, response, n, r2, threshold,
                     per_range, is_key);
  return hipGetLastError();
}

}  // namespace lgicp
