// gicp_shared.h — device helpers shared by the two inner solvers of the GICP loop (gicp.hip:
// Gauss-Newton on the reduced quadratic form; gicp_bfgs.hip: the reference's BFGS on the
// reference's float objective).
#pragma once

#include "host_sincosf.h"
#include "lgicp_device.h"

namespace lgicp {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__device__ __forceinline__ void load_cov6(const double* __restrict__ p, double c[6]) {
  const double2* const q = reinterpret_cast<const double2*>(p);  // 16-byte aligned: 48-byte stride
  const double2 a = q[0], b = q[1], d = q[2];
  c[0] = a.x, c[1] = a.y, c[2] = b.x, c[3] = b.y, c[4] = d.x, c[5] = d.y;
}
// S = R C1 R^T + C2 with C1, C2 as c00 c01 c02 c11 c12 c22 (gicp.hpp:485-493)
__device__ __forceinline__ void sandwich(const double R[9], const double c1[6], const double c2[6],
                                         double& s00, double& s01, double& s02, double& s11,
                                         double& s12, double& s22) {
#pragma clang fp contract(fast)
  double RC[9];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double r0 = R[r * 3], r1 = R[r * 3 + 1], r2 = R[r * 3 + 2];
    RC[r * 3 + 0] = r0 * c1[0] + r1 * c1[1] + r2 * c1[2];
    RC[r * 3 + 1] = r0 * c1[1] + r1 * c1[3] + r2 * c1[4];
    RC[r * 3 + 2] = r0 * c1[2] + r1 * c1[4] + r2 * c1[5];
  }
  s00 = RC[0] * R[0] + RC[1] * R[1] + RC[2] * R[2] + c2[0];
  s01 = RC[0] * R[3] + RC[1] * R[4] + RC[2] * R[5] + c2[1];
  s02 = RC[0] * R[6] + RC[1] * R[7] + RC[2] * R[8] + c2[2];
  s11 = RC[3] * R[3] + RC[4] * R[4] + RC[5] * R[5] + c2[3];
  s12 = RC[3] * R[6] + RC[4] * R[7] + RC[5] * R[8] + c2[4];
  s22 = RC[6] * R[6] + RC[7] * R[7] + RC[8] * R[8] + c2[5];
}

// 1 / x for a positive, normal x: hardware estimate and two Newton steps (an ulp or two off the
// rounded quotient, a third of the instructions of the IEEE division sequence)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}

// applyState on the identity (gicp.hpp:619-634): Eigen multiplies the three AngleAxisf as
// float quaternions and converts the product to a rotation matrix
// HOST_LIBM: the float sine and cosine are the host libm's (host_sincosf.h) instead of the
// device's own, which differ from it in the last bit for a few arguments in a thousand: the BFGS
// back end is compared with a host run evaluation by evaluation.
template <bool HOST_LIBM = false>
static __device__ void apply_state(const double x[6], float T[16]) {
  const float az = (float)x[5], ay = (float)x[4], ax = (float)x[3];
  auto hc = [](float h) { return HOST_LIBM ? host_cosf(h) : cosf(h); };
  auto hs = [](float h) { return HOST_LIBM ? host_sinf(h) : sinf(h); };
  const float qz[4] = {hc(0.5f * az), 0.f, 0.f, hs(0.5f * az)};
  const float qy[4] = {hc(0.5f * ay), 0.f, hs(0.5f * ay), 0.f};
  const float qx[4] = {hc(0.5f * ax), hs(0.5f * ax), 0.f, 0.f};
  float a[4], q[4];
  a[0] = qz[0] * qy[0] - qz[1] * qy[1] - qz[2] * qy[2] - qz[3] * qy[3];
  a[1] = qz[0] * qy[1] + qz[1] * qy[0] + qz[2] * qy[3] - qz[3] * qy[2];
  a[2] = qz[0] * qy[2] + qz[2] * qy[0] + qz[3] * qy[1] - qz[1] * qy[3];
  a[3] = qz[0] * qy[3] + qz[3] * qy[0] + qz[1] * qy[2] - qz[2] * qy[1];
  q[0] = a[0] * qx[0] - a[1] * qx[1] - a[2] * qx[2] - a[3] * qx[3];
  q[1] = a[0] * qx[1] + a[1] * qx[0] + a[2] * qx[3] - a[3] * qx[2];
  q[2] = a[0] * qx[2] + a[2] * qx[0] + a[3] * qx[1] - a[1] * qx[3];
  q[3] = a[0] * qx[3] + a[3] * qx[0] + a[1] * qx[2] - a[2] * qx[1];
  const float w = q[0], X = q[1], Y = q[2], Z = q[3];
  const float tx = 2 * X, ty = 2 * Y, tz = 2 * Z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * X, txy = ty * X, txz = tz * X;
  const float tyy = ty * Y, tyz = tz * Y, tzz = tz * Z;
  for (int i = 0; i < 16; ++i) T[i] = 0.f;
  T[0] = 1 - (tyy + tzz);
  T[1] = txy - twz;
  T[2] = txz + twy;
  T[4] = txy + twz;
  T[5] = 1 - (txx + tzz);
  T[6] = tyz - twx;
  T[8] = txz - twy;
  T[9] = tyz + twx;
  T[10] = 1 - (txx + tyy);
  T[3] = (float)x[0];
  T[7] = (float)x[1];
  T[11] = (float)x[2];
  T[15] = 1.f;
}

// End of a solve workgroup's part in an outer iteration, called by one thread of every workgroup of
// the launch as its last action.  The last workgroup to arrive publishes the iteration number in
// host memory: the host then knows that the solve kernel of that iteration is through, without an
// event between the kernels of the stream.  `ticket` counts arrivals and is back at zero when the
// launch ends.  A thread's earlier stores to host memory (its pair's flag) have left the compute
// unit before it takes its ticket, nothing is written back from the caches for it: should a flag
// become visible to the host after the number all the same, the host takes the pair for active one
// iteration longer, which changes no result (what a pair does is decided by its state on the
// device).
__device__ __forceinline__ void iteration_tick(int* ticket, int n_blocks, int* host_iter, int iter) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (atomicAdd(ticket, 1) == n_blocks - 1) {
    atomicExch(ticket, 0);
    __hip_atomic_store(host_iter, iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static __device__ void mat4f_mul(const float a[16], const float b[16], float out[16]) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float s = a[i * 4 + 0] * b[0 * 4 + j];
      s = s + a[i * 4 + 1] * b[1 * 4 + j];
      s = s + a[i * 4 + 2] * b[2 * 4 + j];
      s = s + a[i * 4 + 3] * b[3 * 4 + j];
      out[i * 4 + j] = s;
    }
}

}  // namespace lgicp
