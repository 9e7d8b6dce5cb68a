// knn_cov.hip — k-nearest-neighbour covariance estimation with plane regularisation.
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

template <int K>
__global__ __launch_bounds__(NN_THREADS) void cov_knn_kernel(const float4* __restrict__ xyz, int n,
                                                             int n_pad, int k, double eps,
                                                             double* __restrict__ cov6) {
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
    hipLaunchKernelGGL(cov_knn_kernel<10>, grid, block, 0, stream, xyz, n, n_pad, k, eps, cov6);
  else if (k <= 20)
    hipLaunchKernelGGL(cov_knn_kernel<20>, grid, block, 0, stream, xyz, n, n_pad, k, eps, cov6);
  else
    hipLaunchKernelGGL(cov_knn_kernel<32>, grid, block, 0, stream, xyz, n, n_pad, k, eps, cov6);
  return hipGetLastError();
}

}  // namespace lgicp
