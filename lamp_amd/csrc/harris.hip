// harris.hip — Harris3D keypoints on the normals of a cloud, for the SAC-IA / TEASER++ front end.
//
// Replaces lamp_utils::ComputeKeypoints with precomputed normals
// (lamp_utils/src/PointCloudUtils.cc:78-96): pcl::HarrisKeypoint3D, HARRIS response, radius
// search, threshold, non-maxima suppression, no refinement.  Response of a point = 0.04 + det(C)
// - 0.04 trace(C)^2 in float, C = mean of n n^T over the neighbours within the radius (finite
// normals only); a point is a keypoint when its response is finite, >= threshold and no
// neighbour within the radius has a strictly larger response.
//
// Mapping: two tiled sweeps (response, then suppression) with the cloud streaming through LDS in
// its own order; the radius test uses the search distance arithmetic of the 1-NN kernel and FLANN's
// strict `<`.  A scan arrives ring by ring, so 64 consecutive points are a short arc: the boxes of
// those chunks (chunk_boxes_kernel) let a workgroup skip the tiles, and a wave the chunks, that lie
// farther from its points than the radius (the float gap to a box never exceeds the float distance
// to a point inside it, so nothing within the radius is skipped and the sums keep their order).  Both sweeps give a wave four points at a time and a lane a candidate (see the
// kernels).  Neighbours are summed
// in ascending index (PCL visits them by ascending distance), so responses can differ from PCL's in
// the last float bits.
#include <algorithm>

#include "lgicp_device.h"

namespace lgicp {

static constexpr int HR_TILE = 512;

// boxes[2 c], boxes[2 c + 1]: smallest and largest finite coordinates of points 64 c .. 64 c + 63
// (+inf / -inf where a chunk has none)
__global__ __launch_bounds__(256) void chunk_boxes_kernel(const float4* __restrict__ xyz, int n,
                                                          float4* __restrict__ boxes) {
  const int lane = threadIdx.x & 63;
  const int chunk = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (chunk * 64 >= n) return;  // wave-uniform
  const int i = chunk * 64 + lane;
  const float inf = __builtin_inff();
  float x0 = inf, y0 = inf, z0 = inf, x1 = -inf, y1 = -inf, z1 = -inf;
  if (i < n) {
    const float4 p = xyz[i];
    // (fminf / fmaxf drop a NaN operand; infinite coordinates stay out as well)
    if (__builtin_isfinite(p.x)) x0 = x1 = p.x;
    if (__builtin_isfinite(p.y)) y0 = y1 = p.y;
    if (__builtin_isfinite(p.z)) z0 = z1 = p.z;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    x0 = __builtin_fminf(x0, __shfl_xor(x0, off)), x1 = __builtin_fmaxf(x1, __shfl_xor(x1, off));
    y0 = __builtin_fminf(y0, __shfl_xor(y0, off)), y1 = __builtin_fmaxf(y1, __shfl_xor(y1, off));
    z0 = __builtin_fminf(z0, __shfl_xor(z0, off)), z1 = __builtin_fmaxf(z1, __shfl_xor(z1, off));
  }
  if (lane == 0) {
    boxes[2 * chunk] = make_float4(x0, y0, z0, 0.f);
    boxes[2 * chunk + 1] = make_float4(x1, y1, z1, 0.f);
  }
}

hipError_t launch_chunk_boxes(const float4* xyz, int n, float4* boxes, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(chunk_boxes_kernel, dim3(((n + 63) / 64 + 3) / 4), dim3(256), 0, stream, xyz, n, boxes);
  return hipGetLastError();
}

// One lane per candidate, not per point: a wave owns HR_PW points at a time, its 64 lanes test 64
// candidates of the tile in LDS against each of them, and the few that lie within the radius (a
// ballot) are added to the point's sums one after the other in ascending index - the order the
// sums had with one lane per point, so the responses are the same floats, but a 16k-point cloud is
// 1,024 workgroups instead of 64.  The seven sums of a point sit in seven lanes (lane 0: n_x n_x,
// 1: n_y n_x, 2: n_z n_x, 3: n_x n_y, 4: n_y n_y, 5: n_z n_y, 6: n_z n_z): a neighbour costs one
// multiplication and one addition.  Its normal is not fetched from LDS when its turn comes (two
// dependent round trips per neighbour were most of the kernel's time): every lane holds the normal
// of its own candidate, read once per chunk that has a neighbour at all, and a neighbour's three
// components are broadcast out of its lane, each sum's lane picking its two factors.
static constexpr int HR_PW = 4;
__device__ __forceinline__ float bcast(float v, int l) {  // lane l's value, l wave-uniform
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), l));
}
__global__ __launch_bounds__(256) void harris_response_kernel(const float4* __restrict__ xyz,
                                                              const float4* __restrict__ nrm,
                                                              int n, float r2,
                                                              const float4* __restrict__ boxes,
                                                              float* __restrict__ response) {
  __shared__ __attribute__((aligned(16))) float tile[6 * HR_TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = (blockIdx.x * 4 + wave) * HR_PW;  // this wave's points: i0 .. i0 + HR_PW - 1
  float px[HR_PW], py[HR_PW], pz[HR_PW];
  float sum[HR_PW];  // lane L < 7: sum L of the point
  unsigned count[HR_PW];
  // the components of a neighbour's normal (0: n_x, 1: n_y, 2: n_z) this lane's factors are
  const int comp_a = lane < 6 ? lane % 3 : 2, comp_b = lane < 3 ? 0 : lane < 6 ? 1 : 2;
  const bool a_x = comp_a == 0, a_y = comp_a == 1, b_x = comp_b == 0, b_y = comp_b == 1;
#pragma unroll
  for (int q = 0; q < HR_PW; ++q) {
    float4 p = make_float4(__builtin_nanf(""), 0, 0, 0);
    if (i0 + q < n) p = xyz[i0 + q];
    px[q] = p.x, py[q] = p.y, pz[q] = p.z;
    sum[q] = 0.f;
    count[q] = 0;
  }
  for (int base = 0; base < n; base += HR_TILE) {
    unsigned reach[HR_PW], reach_any = 0;
#pragma unroll
    for (int q = 0; q < HR_PW; ++q) {
      reach[q] = chunks_in_reach(boxes, n, base, HR_TILE / 64, lane, px[q], py[q], pz[q], r2);
      reach_any |= reach[q];
    }
    // (also the barrier that keeps the previous tile in place until every wave is done with it)
    if (!__syncthreads_or(reach_any != 0)) continue;
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
    const int jend = min(HR_TILE, n - base);  // (the rest of the tile is padding, never within the radius)
    for (int c = 0; c < jend; c += 64) {
      const unsigned bit = 1u << (c >> 6);
      if (!(reach_any & bit)) continue;  // wave-uniform
      const float tx = tile[c + lane], ty = tile[HR_TILE + c + lane], tz = tile[2 * HR_TILE + c + lane];
      // The wave's points follow each other in the scan and share most of their neighbours: the
      // candidates within the radius of any of them are gone through once, in ascending order, a
      // candidate's products are formed once and added to the sums of the points it is a neighbour
      // of - every point still receives its neighbours in ascending index.
      unsigned long long mq[HR_PW], any = 0;
#pragma unroll
      for (int q = 0; q < HR_PW; ++q) {
        mq[q] = 0;
        if (!(reach[q] & bit)) continue;  // wave-uniform
        const float dx = px[q] - tx, dy = py[q] - ty, dz = pz[q] - tz;
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        mq[q] = __ballot(d2 < r2);
        any |= mq[q];
      }
      if (!any) continue;
      // this lane's candidate's normal
      const float cnx = tile[3 * HR_TILE + c + lane], cny = tile[4 * HR_TILE + c + lane], cnz = tile[5 * HR_TILE + c + lane];
      while (any) {  // wave-uniform
        const int l = __builtin_ctzll(any);
        any &= any - 1;
        const float sx = bcast(cnx, l), sy = bcast(cny, l), sz = bcast(cnz, l);
        const bool counts = isfinite(sx);  // a neighbour without a normal does not count
        const float fa = a_x ? sx : (a_y ? sy : sz), fb = b_x ? sx : (b_y ? sy : sz);
        const float prod = fa * fb;
        // No branch per point: a point the candidate is no neighbour of adds +0, which leaves a
        // sum as it is (the sums start at +0 and a sum of floats that started at +0 is never -0).
#pragma unroll
        for (int q = 0; q < HR_PW; ++q) {
          const bool take = counts && ((mq[q] >> l) & 1ull);  // wave-uniform
          sum[q] += take ? prod : 0.f;
          count[q] += take ? 1u : 0u;
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < HR_PW; ++q) {
    const int i = i0 + q;
    if (i >= n) break;
    float r = 0.f;
    auto of_lane = [&](int l) {
      return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(sum[q]), l));
    };
    float sxx = of_lane(0), sxy = of_lane(1), sxz = of_lane(2), syy = of_lane(4), syz = of_lane(5), szz = of_lane(6);
    if (isfinite(px[q]) && isfinite(py[q]) && isfinite(pz[q])) {
      if (count[q] > 0) {
        const float cf = (float)count[q];
        sxx = sxx / cf, sxy = sxy / cf, sxz = sxz / cf, syy = syy / cf, syz = syz / cf, szz = szz / cf;
      }
      const float trace = sxx + syy + szz;
      if (trace != 0) {
        const float det = sxx * syy * szz + 2.0f * sxy * sxz * syz - sxz * sxz * syy - sxy * sxy * szz - syz * syz * sxx;
        r = 0.04f + det - 0.04f * trace * trace;
      }
    }
    if (lane == 0) response[i] = r;
  }
}

// Suppression, one lane per candidate like the response sweep: a wave owns HR_PW points at a time,
// its 64 lanes test 64 candidates of the tile against each, and a point is done with as soon as one
// neighbour with a strictly larger response has shown up (most points are, early in the sweep); a
// wave whose points are all done only keeps the workgroup's barriers company.
__global__ __launch_bounds__(256) void harris_nms_kernel(const float4* __restrict__ xyz,
                                                         const float* __restrict__ response,
                                                         int n, float r2, float threshold,
                                                         const float4* __restrict__ boxes,
                                                         unsigned char* __restrict__ is_key) {
  __shared__ __attribute__((aligned(16))) float tile[4 * HR_TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = (blockIdx.x * 4 + wave) * HR_PW;
  float px[HR_PW], py[HR_PW], pz[HR_PW], ri[HR_PW];
  bool cand[HR_PW];
  bool any = false;
#pragma unroll
  for (int q = 0; q < HR_PW; ++q) {
    float4 p = make_float4(__builtin_nanf(""), 0, 0, 0);
    ri[q] = -1.f;
    if (i0 + q < n) p = xyz[i0 + q], ri[q] = response[i0 + q];
    px[q] = p.x, py[q] = p.y, pz[q] = p.z;
    cand[q] = i0 + q < n && isfinite(p.x) && isfinite(p.y) && isfinite(p.z) && isfinite(ri[q]) &&
              !(ri[q] < threshold);
    any = any || cand[q];
  }
  const bool block_any = __syncthreads_count(any) != 0;
  for (int base = 0; block_any && base < n; base += HR_TILE) {
    unsigned reach[HR_PW], reach_any = 0;
#pragma unroll
    for (int q = 0; q < HR_PW; ++q) {
      reach[q] = cand[q] ? chunks_in_reach(boxes, n, base, HR_TILE / 64, lane, px[q], py[q], pz[q], r2) : 0u;
      reach_any |= reach[q];
    }
    if (!__syncthreads_or(reach_any != 0)) continue;  // (also the barrier between two tiles)
    for (int j = tid; j < HR_TILE; j += 256) {
      float4 q = make_float4(PAD_COORD, PAD_COORD, PAD_COORD, 0);
      float rj = -1.f;
      if (base + j < n) {
        q = xyz[base + j];
        rj = response[base + j];
      }
      tile[j] = q.x;
      tile[HR_TILE + j] = q.y;
      tile[2 * HR_TILE + j] = q.z;
      tile[3 * HR_TILE + j] = rj;
    }
    __syncthreads();
    if (!(cand[0] || cand[1] || cand[2] || cand[3])) continue;  // wave-uniform
    const int jend = min(HR_TILE, n - base);  // (the rest of the tile is padding, never within the radius)
    for (int c = 0; c < jend; c += 64) {
      const unsigned bit = 1u << (c >> 6);
      if (!(reach_any & bit)) continue;  // wave-uniform
      const float tx = tile[c + lane], ty = tile[HR_TILE + c + lane], tz = tile[2 * HR_TILE + c + lane];
      const float rj = tile[3 * HR_TILE + c + lane];
#pragma unroll
      for (int q = 0; q < HR_PW; ++q) {
        if (!cand[q] || !(reach[q] & bit)) continue;  // wave-uniform
        const float dx = px[q] - tx, dy = py[q] - ty, dz = pz[q] - tz;
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        if (__ballot(d2 < r2 && ri[q] < rj)) cand[q] = false;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < HR_PW; ++q)
    if (lane == 0 && i0 + q < n) is_key[i0 + q] = cand[q] ? 1 : 0;
}

hipError_t launch_harris(const float4* xyz, const float4* nrm, int n, float r2, float threshold,
                         float* response, unsigned char* is_key, const float4* boxes,
                         hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const dim3 block(256);
  hipLaunchKernelGGL(harris_response_kernel, dim3((n + 4 * HR_PW - 1) / (4 * HR_PW)), block, 0, stream, xyz, nrm, n, r2,
                     boxes, response);
  hipLaunchKernelGGL(harris_nms_kernel, dim3((n + 4 * HR_PW - 1) / (4 * HR_PW)), block, 0, stream, xyz, response, n, r2,
                     threshold, boxes, is_key);
  return hipGetLastError();
}

}  // namespace lgicp
