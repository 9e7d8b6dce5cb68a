// knn_cov.hip — k-nearest-neighbour covariance estimation with plane regularisation, and the
// k-nearest-neighbour surface normals of lamp_utils::ComputeNormals (same neighbour search).
//
// Replaces the k-NN branch of MultithreadedGICP::computeCovariances
// (lamp_utils/include/lamp_utils/gicp.hpp:83-155): for every point the k nearest points of
// its own cloud (the point itself included), their mean and second moments accumulated in
// double from float products and divided by k, and the covariance rebuilt from its singular
// vectors with singular values (1, 1, gicp_epsilon).
//
// Mapping: one lane per query point; the cloud streams through LDS tiles exactly as in the
// 1-NN kernel (same unfused distance arithmetic); each lane keeps its k best (distance, index)
// pairs sorted in registers and only enters the compare-exchange chain when a candidate beats
// its current worst, which becomes rare after the first few tiles.  Because
// U diag(1,1,eps) U^T = I - (1-eps) u3 u3^T only the singular vector of the smallest singular
// value is needed; it comes from a cyclic Jacobi iteration on the symmetric 3x3.
#include "lgicp_device.h"

namespace lgicp {

// pcl::computeRoots2: roots of the quadratic left when one eigenvalue is zero
__device__ __forceinline__ void roots2(float b, float c, float roots[3]) {
  roots[0] = 0.f;
  float d = b * b - 4.0f * c;
  if (d < 0.0f) d = 0.0f;
  const float sd = sqrtf(d);
  roots[2] = 0.5f * (b + sd);
  roots[1] = 0.5f * (b - sd);
}

// pcl::computeRoots: eigenvalues of a symmetric 3x3 (row-major m) in closed form, ascending
__device__ void compute_roots(const float m[9], float roots[3]) {
  const float c0 = m[0] * m[4] * m[8] + 2.0f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] -
                   m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
  const float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
  const float c2 = m[0] + m[4] + m[8];
  if (fabsf(c0) < 1.1920929e-07f) {
    roots2(c2, c1, roots);
    return;
  }
  const float s_inv3 = 1.0f / 3.0f;
  const float s_sqrt3 = sqrtf(3.0f);
  const float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.0f) a_over_3 = 0.0f;
  const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
  float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.0f) q = 0.0f;
  const float rho = sqrtf(-a_over_3);
  const float theta = atan2f(sqrtf(-q), half_b) * s_inv3;
  const float cos_theta = cosf(theta), sin_theta = sinf(theta);
  roots[0] = c2_over_3 + 2.0f * rho * cos_theta;
  roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  float t;
  if (roots[0] >= roots[1]) t = roots[0], roots[0] = roots[1], roots[1] = t;
  if (roots[1] >= roots[2]) {
    t = roots[1], roots[1] = roots[2], roots[2] = t;
    if (roots[0] >= roots[1]) t = roots[0], roots[0] = roots[1], roots[1] = t;
  }
  if (roots[0] <= 0.0f) roots2(c2, c1, roots);
}

// pcl::NormalEstimation::computePointNormal from the single-pass float sums of `count` points
// (xx xy xz yy yz zz x y z): covariance, smallest eigenvector by the closed-form roots + row cross
// products of pcl::eigen33, curvature, flip towards the viewpoint (0, 0, 0).  Returns
// (nx, ny, nz, curvature).
__device__ float4 normal_from_moments(float accu[9], int count, float qx, float qy, float qz) {
  const float fk = (float)count;
#pragma unroll
  for (int e = 0; e < 9; ++e) accu[e] /= fk;
  float cov[9];
  cov[0] = accu[0] - accu[6] * accu[6];
  cov[1] = cov[3] = accu[1] - accu[6] * accu[7];
  cov[2] = cov[6] = accu[2] - accu[6] * accu[8];
  cov[4] = accu[3] - accu[7] * accu[7];
  cov[5] = cov[7] = accu[4] - accu[7] * accu[8];
  cov[8] = accu[5] - accu[8] * accu[8];
  float scale = 0.f;
#pragma unroll
  for (int e = 0; e < 9; ++e) scale = fabsf(cov[e]) > scale ? fabsf(cov[e]) : scale;
  if (scale <= 1.17549435e-38f) scale = 1.0f;
  float sm[9];
#pragma unroll
  for (int e = 0; e < 9; ++e) sm[e] = cov[e] / scale;
  float roots[3];
  compute_roots(sm, roots);
  const float ev = roots[0] * scale;
  sm[0] -= roots[0];
  sm[4] -= roots[0];
  sm[8] -= roots[0];
  const float v1[3] = {sm[1] * sm[5] - sm[2] * sm[4], sm[2] * sm[3] - sm[0] * sm[5], sm[0] * sm[4] - sm[1] * sm[3]};
  const float v2[3] = {sm[1] * sm[8] - sm[2] * sm[7], sm[2] * sm[6] - sm[0] * sm[8], sm[0] * sm[7] - sm[1] * sm[6]};
  const float v3[3] = {sm[4] * sm[8] - sm[5] * sm[7], sm[5] * sm[6] - sm[3] * sm[8], sm[3] * sm[7] - sm[4] * sm[6]};
  const float l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
  const float l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
  const float l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
  float nx, ny, nz, len;
  if (l1 >= l2 && l1 >= l3)
    nx = v1[0], ny = v1[1], nz = v1[2], len = l1;
  else if (l2 >= l1 && l2 >= l3)
    nx = v2[0], ny = v2[1], nz = v2[2], len = l2;
  else
    nx = v3[0], ny = v3[1], nz = v3[2], len = l3;
  const float r = sqrtf(len);
  nx /= r, ny /= r, nz /= r;
  const float eig_sum = cov[0] + cov[4] + cov[8];
  const float curv = eig_sum != 0.f ? fabsf(ev / eig_sum) : 0.f;
  const float vx = 0.f - qx, vy = 0.f - qy, vz = 0.f - qz;
  const float cos_theta = vx * nx + vy * ny + vz * nz;
  if (cos_theta < 0.f) nx *= -1.f, ny *= -1.f, nz *= -1.f;
  return make_float4(nx, ny, nz, curv);
}

// The normals pcl::HarrisKeypoint3D estimates for itself when none are handed in (the
// lamp_utils::ComputeKeypoints overload without normals, lamp_utils/src/PointCloudUtils.cc:61-76):
// pcl::NormalEstimation over a radius search.  One lane per point, the cloud streaming through
// LDS; neighbours = points at squared distance < r2 (FLANN's strict test, the point itself
// included) summed in ascending index (PCL visits them by ascending distance, so the last float
// bits can differ); fewer than 3 neighbours give a NaN normal.
static constexpr int NR_TILE = 512;
__global__ __launch_bounds__(256) void normals_radius_kernel(const float4* __restrict__ xyz, int n,
                                                             float r2, float4* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float tile[3 * NR_TILE];
  const int tid = threadIdx.x;
  const int i = blockIdx.x * 256 + tid;
  float4 p = make_float4(__builtin_nanf(""), 0, 0, 0);
  if (i < n) p = xyz[i];
  float accu[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int count = 0;
  for (int base = 0; base < n; base += NR_TILE) {
    __syncthreads();
    for (int j = tid; j < NR_TILE; j += 256) {
      float4 q = make_float4(PAD_COORD, PAD_COORD, PAD_COORD, 0);
      if (base + j < n) q = xyz[base + j];
      tile[j] = q.x;
      tile[NR_TILE + j] = q.y;
      tile[2 * NR_TILE + j] = q.z;
    }
    __syncthreads();
    const int jend = min(NR_TILE, n - base);
    for (int j = 0; j < jend; ++j) {
      const float x = tile[j], y = tile[NR_TILE + j], z = tile[2 * NR_TILE + j];
      const float dx = p.x - x, dy = p.y - y, dz = p.z - z;
      const float d2 = (dx * dx + dy * dy) + dz * dz;
      if (!(d2 < r2)) continue;
      accu[0] += x * x;
      accu[1] += x * y;
      accu[2] += x * z;
      accu[3] += y * y;
      accu[4] += y * z;
      accu[5] += z * z;
      accu[6] += x;
      accu[7] += y;
      accu[8] += z;
      ++count;
    }
  }
  if (i >= n) return;
  const float nan = __builtin_nanf("");
  float4 res = make_float4(nan, nan, nan, nan);
  if (count >= 3 && isfinite(p.x) && isfinite(p.y) && isfinite(p.z))
    res = normal_from_moments(accu, count, p.x, p.y, p.z);
  out[i] = res;
}

hipError_t launch_normals_radius(const float4* xyz, int n, float r2, float4* normals_curv,
                                 hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(normals_radius_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, xyz, n, r2,
                     normals_curv);
  return hipGetLastError();
}

// MODE 0: GICP covariance (gicp.hpp:83-155) into out_d[n][6].
// MODE 1: surface normal + curvature of pcl::NormalEstimation (lamp_utils::ComputeNormals,
//         lamp_utils/include/lamp_utils/PointCloudUtils.h:38-62) into out_f[n] = (nx, ny, nz, curvature):
//         single-pass float moments of the k neighbours in ascending distance order, smallest
//         eigenvector by the closed-form roots + row cross products of pcl::eigen33, flipped
//         towards the viewpoint (0, 0, 0).
template <int K, int MODE>
__global__ __launch_bounds__(NN_THREADS) void cov_knn_kernel(const float4* __restrict__ xyz, int n,
                                                             int n_pad, int k, double eps,
                                                             double* __restrict__ cov6,
                                                             float4* __restrict__ out_f) {
  __shared__ __attribute__((aligned(16))) float lds[3 * NN_TILE];
  float* const lx = lds;
  float* const ly = lds + NN_TILE;
  float* const lz = lds + 2 * NN_TILE;
  const int tid = threadIdx.x;
  const int i = blockIdx.x * NN_THREADS + tid;
  float qx = __builtin_nanf(""), qy = qx, qz = qx;
  if (i < n) {
    const float4 p = xyz[i];
    qx = p.x;
    qy = p.y;
    qz = p.z;
  }
  float bd[K];
  int bi[K];
#pragma unroll
  for (int s = 0; s < K; ++s) {
    bd[s] = __builtin_inff();
    bi[s] = 0x7FFFFFFF;
  }
  // slots >= k are never filled: the admission test looks at slot k-1
  const int n_tiles = n_pad / NN_TILE;
  for (int tile = 0; tile < n_tiles; ++tile) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NN_TILE / NN_THREADS; ++r) {
      const int j = r * NN_THREADS + tid;
      const float4 p = xyz[(size_t)tile * NN_TILE + j];
      lx[j] = p.x;
      ly[j] = p.y;
      lz[j] = p.z;
    }
    __syncthreads();
    const int jend = min(NN_TILE, n - tile * NN_TILE);
    for (int j = 0; j < jend; ++j) {
      const float dx = qx - lx[j], dy = qy - ly[j], dz = qz - lz[j];
      const float d = (dx * dx + dy * dy) + dz * dz;
      const int gj = tile * NN_TILE + j;
      float worst = bd[0];
      int worst_i = bi[0];
#pragma unroll
      for (int s = 1; s < K; ++s)
        if (s < k) {
          worst = bd[s];
          worst_i = bi[s];
        }
      if (d < worst || (d == worst && gj < worst_i)) {
        float cd = d;
        int ci = gj;
#pragma unroll
        for (int s = 0; s < K; ++s) {
          if (s < k) {
            const bool lt = cd < bd[s] || (cd == bd[s] && ci < bi[s]);
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
  }
  if (i >= n) return;

  if (MODE == 1) {
    float4 res = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
    if (k >= 3) {
      float accu[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < K; ++s) {
        if (s < k && bi[s] != 0x7FFFFFFF) {
          const float4 p = xyz[bi[s]];
          accu[0] += p.x * p.x;
          accu[1] += p.x * p.y;
          accu[2] += p.x * p.z;
          accu[3] += p.y * p.y;
          accu[4] += p.y * p.z;
          accu[5] += p.z * p.z;
          accu[6] += p.x;
          accu[7] += p.y;
          accu[8] += p.z;
        }
      }
      res = normal_from_moments(accu, k, qx, qy, qz);
    }
    out_f[i] = res;
    return;
  }

  double mean[3] = {0, 0, 0}, c00 = 0, c10 = 0, c11 = 0, c20 = 0, c21 = 0, c22 = 0;
#pragma unroll
  for (int s = 0; s < K; ++s) {
    if (s < k && bi[s] != 0x7FFFFFFF) {
      const float4 p = xyz[bi[s]];
      mean[0] += p.x;
      mean[1] += p.y;
      mean[2] += p.z;
      c00 += (double)(p.x * p.x);
      c10 += (double)(p.y * p.x);
      c11 += (double)(p.y * p.y);
      c20 += (double)(p.z * p.x);
      c21 += (double)(p.z * p.y);
      c22 += (double)(p.z * p.z);
    }
  }
  const double kk = (double)k;
  mean[0] /= kk;
  mean[1] /= kk;
  mean[2] /= kk;
  double A[9];
  A[0] = c00 / kk - mean[0] * mean[0];
  A[3] = A[1] = c10 / kk - mean[1] * mean[0];
  A[4] = c11 / kk - mean[1] * mean[1];
  A[6] = A[2] = c20 / kk - mean[2] * mean[0];
  A[7] = A[5] = c21 / kk - mean[2] * mean[1];
  A[8] = c22 / kk - mean[2] * mean[2];
  double V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 32; ++sweep) {
    const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    const double diag = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off == 0.0 || 2 * off <= 1e-32 * (diag + 2 * off)) break;
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      const double apq = A[p * 3 + q];
      if (apq == 0.0) continue;
      const double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
      const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double arp = A[r * 3 + p], arq = A[r * 3 + q];
        A[r * 3 + p] = c * arp - s * arq;
        A[r * 3 + q] = s * arp + c * arq;
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double apr = A[p * 3 + r], aqr = A[q * 3 + r];
        A[p * 3 + r] = c * apr - s * aqr;
        A[q * 3 + r] = s * apr + c * aqr;
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double vrp = V[r * 3 + p], vrq = V[r * 3 + q];
        V[r * 3 + p] = c * vrp - s * vrq;
        V[r * 3 + q] = s * vrp + c * vrq;
      }
    }
  }
  int smallest = 0;
  if (A[4] < A[smallest * 4]) smallest = 1;
  if (A[8] < A[smallest * 4]) smallest = 2;
  double u0 = V[0 + smallest], u1 = V[3 + smallest], u2 = V[6 + smallest];
  const double w = 1.0 - eps;
  double* c = cov6 + 6 * (size_t)i;
  c[0] = 1.0 - w * u0 * u0;
  c[1] = -w * u0 * u1;
  c[2] = -w * u0 * u2;
  c[3] = 1.0 - w * u1 * u1;
  c[4] = -w * u1 * u2;
  c[5] = 1.0 - w * u2 * u2;
}

hipError_t launch_cov_knn(const float4* xyz, int n, int n_pad, int k, double eps, double* cov6,
                          hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  if (k < 1 || k > 32) return hipErrorInvalidValue;
  const dim3 grid((n + NN_THREADS - 1) / NN_THREADS), block(NN_THREADS);
  if (k <= 10)
    hipLaunchKernelGGL((cov_knn_kernel<10, 0>), grid, block, 0, stream, xyz, n, n_pad, k, eps, cov6, nullptr);
  else if (k <= 20)
    hipLaunchKernelGGL((cov_knn_kernel<20, 0>), grid, block, 0, stream, xyz, n, n_pad, k, eps, cov6, nullptr);
  else
    hipLaunchKernelGGL((cov_knn_kernel<32, 0>), grid, block, 0, stream, xyz, n, n_pad, k, eps, cov6, nullptr);
  return hipGetLastError();
}

hipError_t launch_normals_knn(const float4* xyz, int n, int n_pad, int k, float4* normals_curv,
                              hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  if (k < 1 || k > 32) return hipErrorInvalidValue;
  const dim3 grid((n + NN_THREADS - 1) / NN_THREADS), block(NN_THREADS);
  if (k <= 10)
    hipLaunchKernelGGL((cov_knn_kernel<10, 1>), grid, block, 0, stream, xyz, n, n_pad, k, 0.0, nullptr, normals_curv);
  else if (k <= 20)
    hipLaunchKernelGGL((cov_knn_kernel<20, 1>), grid, block, 0, stream, xyz, n, n_pad, k, 0.0, nullptr, normals_curv);
  else
    hipLaunchKernelGGL((cov_knn_kernel<32, 1>), grid, block, 0, stream, xyz, n, n_pad, k, 0.0, nullptr, normals_curv);
  return hipGetLastError();
}

}  // namespace lgicp
