// gicp.hip — Mahalanobis linearisation/reduction, on-device Gauss-Newton solve and SE(3)
// update, and the post-alignment reductions (fitness, Ap) for gfx950.
//
// Reference being replaced: the serial O(m) cost/gradient loops the BFGS optimiser calls
// (lamp_utils/include/lamp_utils/gicp.hpp:290-402), the Mahalanobis build inside the lookup
// loop (gicp.hpp:483-497), estimateRigidTransformationBFGS (gicp.hpp:217-287), the
// convergence test (gicp.hpp:526-541,564-580), pcl getFitnessScore and
// ComputeAp_ForPoint2PlaneICP (lamp_utils/src/PointCloudUtils.cc:115-163).
//
// The residual res = Theta*(p,1) - q is linear in the 12 entries of Theta = [R|t], so the
// objective (1/m) sum res^T M res of one outer iteration is a quadratic form in
// delta = vec(Theta - Theta0).  One pass over the correspondences reduces it to 74 doubles
// per pair (f64 matrix instructions within a wave, four waves per slab of source points);
// the Gauss-Newton iterations on the 6 Euler parameters then cost O(1) each and run in the
// solve kernel, which also applies the reference's convergence test and produces the next
// float transformation with the reference's applyState arithmetic (gicp.hpp:619-634).
#include "lgicp_device.h"
#include "gicp_shared.h"

namespace lgicp {

// second-moment slot of (k,l), k <= l, in the order xx xy xz x1 yy yz y1 zz z1 11
__host__ __device__ constexpr int uu_slot(int k, int l) {
  return k == 0 ? l : (k == 1 ? 3 + l : (k == 2 ? 5 + l : 9));
}

// ---------------------------------------------------------------------------------------------
// The statistics are sums over the correspondences of outer products: with
//   a = (xx xy xz x yy yz y zz z 1 | r0 r1 r2)   (u u^T in uu_slot order, then the residual)
//   b = (M00 M01 M02 M11 M12 M22 | Mr0 Mr1 Mr2 | 1)
// Q is rows 0..9 x columns 0..5 of D = sum a b^T, the linear term rows (x, y, z, 1) x columns
// 6..8, the constant D[10][6] + D[11][7] + D[12][8] and the count D[9][9].  That is a
// (13 x m)(m x 10) matrix product with the correspondences as the contraction index, which the
// f64 matrix pipe does as v_mfma_f64_16x16x4_f64: four correspondences per instruction, the
// whole 16 x 16 accumulator in four f64 registers per lane - against 73 f64 accumulators per lane
// when every lane sums its own points, which capped that version at two waves per SIMD while
// it waited on its gathers.  Each lane still builds a and b of its own point; a wave passes them
// through its private LDS rows so that lane l can feed A[l & 15][l >> 4] and B[l >> 4][l & 15].
typedef double red_d4 __attribute__((ext_vector_type(4)));

constexpr int RM_ROWS_A = 13, RM_ROWS_B = 10;
constexpr int RM_STR = 65;  // doubles per staged row (64 points + 1: operand reads spread over the banks)

// COMPACT: every cloud of the launch carries flagged normals and 32-byte records (covariances
// from normals); otherwise stored covariances and 64-byte records (covariances from k-NN)
template <bool COMPACT>
__global__ __launch_bounds__(RED_THREADS, 3) void gicp_reduce_kernel(
    const CloudDev* __restrict__ clouds, const PairDev* __restrict__ pairs,
    const PairState* __restrict__ st, const int4* __restrict__ bmap, double corr_dist2,
    double* __restrict__ partials) {
  // double-precision statistics: no float result of the reference is reproduced bit by bit
  // here, so multiply-adds may fuse (the float transformation arithmetic lives elsewhere)
#pragma clang fp contract(fast)
  __shared__ double stage[RED_THREADS / 64][(RM_ROWS_A + RM_ROWS_B) * RM_STR];
  const int4 bm = bmap[blockIdx.x];
  const int pair = bm.x;
  const PairState& S = st[pair];
  if (!S.active) return;
  const PairDev& P = pairs[pair];
  const CloudDev& SC = clouds[P.src_cloud];
  const CloudDev& TC = clouds[P.tgt_cloud];

  // R = rotation of transformation_ * guess in double (gicp.hpp:450-460); Theta0 = T widened
  double R[9], Th[12];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += (double)S.T[i * 4 + k] * (double)P.guess[k * 4 + j];
      R[i * 3 + j] = s;
    }
#pragma unroll
  for (int i = 0; i < 12; ++i) Th[i] = (double)S.T[i];  // rows 0..2 of the 4x4, row-major
  // R R^T + 1 (R comes from a float matrix and is orthogonal to 1e-7 only, so this is not 2)
  const double G[6] = {R[0] * R[0] + R[1] * R[1] + R[2] * R[2] + 1.0,
                       R[0] * R[3] + R[1] * R[4] + R[2] * R[5],
                       R[0] * R[6] + R[1] * R[7] + R[2] * R[8],
                       R[3] * R[3] + R[4] * R[4] + R[5] * R[5] + 1.0,
                       R[3] * R[6] + R[4] * R[7] + R[5] * R[8],
                       R[6] * R[6] + R[7] * R[7] + R[8] * R[8] + 1.0};

  const float4* __restrict__ const out = P.out;
  const nnkey_t* __restrict__ const keys = P.keys;
  const float4* __restrict__ const s_cnrm = SC.cnrm;
  const double* __restrict__ const s_cov = SC.cov;
  const float4* __restrict__ const t_rec = TC.rec;
  const double* __restrict__ const t_cov = TC.cov;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* const sa = stage[wave];
  double* const sb = sa + RM_ROWS_A * RM_STR;
  // operand slots of this lane; rows/columns past the used ones only feed unused outputs
  const int orow = min(lane & 15, RM_ROWS_A - 1), ocol = min(lane & 15, RM_ROWS_B - 1);
  const int ok = lane >> 4;
  const double* const pa = sa + orow * RM_STR + ok;
  const double* const pb = sb + ocol * RM_STR + ok;
  red_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  double av[RM_ROWS_A];
#pragma unroll
  for (int r = 0; r < RM_ROWS_A; ++r) av[r] = 0.0;

  const int n = P.n_src;
  const int per = (n + bm.z - 1) / bm.z;
  const int i0 = bm.y * per;
  const int i1 = min(n, i0 + per);
  // One round of 64 points, given S = R C1 R^T + C2 of every lane that holds a correspondence: M,
  // residual, the two operand vectors, staging and the matrix instructions.  A lane without a
  // correspondence contributes b = 0; its a keeps the (finite) values of an earlier round, whose
  // products with zero add nothing.
  auto accumulate = [&](bool valid, const float4 pf, const float4 qf, double s00, double s01,
                        double s02, double s11, double s12, double s22) {
    double bv[RM_ROWS_B];
#pragma unroll
    for (int c = 0; c < RM_ROWS_B; ++c) bv[c] = 0.0;
    if (valid) {
      // M = S^-1 by cofactors (Eigen's 3x3 inverse, gicp.hpp:493)
      const double k00 = s11 * s22 - s12 * s12;
      const double k01 = s02 * s12 - s01 * s22;
      const double k02 = s01 * s12 - s02 * s11;
      const double det = s00 * k00 + s01 * k01 + s02 * k02;
      const double inv = 1.0 / det;
      bv[0] = k00 * inv;
      bv[1] = k01 * inv;
      bv[2] = k02 * inv;
      bv[3] = (s00 * s22 - s02 * s02) * inv;
      bv[4] = (s01 * s02 - s00 * s12) * inv;
      bv[5] = (s00 * s11 - s01 * s01) * inv;
      const double x = (double)pf.x, y = (double)pf.y, z = (double)pf.z;
      const double r0 = (Th[0] * x + Th[1] * y + Th[2] * z + Th[3]) - (double)qf.x;
      const double r1 = (Th[4] * x + Th[5] * y + Th[6] * z + Th[7]) - (double)qf.y;
      const double r2 = (Th[8] * x + Th[9] * y + Th[10] * z + Th[11]) - (double)qf.z;
      bv[6] = bv[0] * r0 + bv[1] * r1 + bv[2] * r2;
      bv[7] = bv[1] * r0 + bv[3] * r1 + bv[4] * r2;
      bv[8] = bv[2] * r0 + bv[4] * r1 + bv[5] * r2;
      bv[9] = 1.0;
      av[0] = x * x, av[1] = x * y, av[2] = x * z, av[3] = x;
      av[4] = y * y, av[5] = y * z, av[6] = y;
      av[7] = z * z, av[8] = z, av[9] = 1.0;
      av[10] = r0, av[11] = r1, av[12] = r2;
    }
#pragma unroll
    for (int r = 0; r < RM_ROWS_A; ++r) sa[r * RM_STR + lane] = av[r];
#pragma unroll
    for (int c = 0; c < RM_ROWS_B; ++c) sb[c * RM_STR + lane] = bv[c];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int t = 0; t < 16; t += 2) {
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * t], pb[4 * t], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * t + 4], pb[4 * t + 4], acc1, 0, 0, 0);
    }
    // the next round's stores must not overtake these reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  // The matches of several rounds are fetched ahead: a round then waits for one memory latency
  // (its gathers) instead of two in a row (match, then gathers).
  constexpr int AHEAD = 4;
  const int first = i0 + wave * 64;
  if constexpr (COMPACT) {
    // Compact clouds keep the stored normal instead of the six doubles: 16 bytes read per source
    // point instead of 48, one 32-byte record gathered per match instead of 64.  With
    // C = 1 - k n n^T (k = 0.999 / |n|^2, see cov_of_regular_normal) the sum is
    // S = (R R^T + 1) - k1 (R n1)(R n1)^T - k2 n2 n2^T: the same quantity as R C1 R^T + C2 whatever
    // R is, in a third of the operations.  The rare points whose normal falls into one of the
    // reference's special branches carry a flag; a wave that holds one redoes those lanes from
    // the stored matrices (wave-uniform test, so the common path has no branch between its loads
    // and their use).
    // The loop is pipelined by hand: the four gathers of a round are issued before the round
    // ahead of it computes (a round is one memory round trip of 1-2 us under load followed by
    // ~0.4 us of arithmetic; with one round in flight the two overlap), the matches AHEAD rounds
    // before they are needed.
    struct Gathered {
      float4 pf, n1, qf, n2;
      int i, j;
      bool valid;
    };
    auto gather = [&](nnkey_t key, int i) {
      Gathered g;
      const float d2 = __uint_as_float((unsigned)(key >> 32));
      g.valid = key != NNKEY_NONE && (double)d2 < corr_dist2;  // gicp.hpp:483
      g.i = i, g.j = (int)(key & 0xFFFFFFFFu);
      g.pf = g.n1 = g.qf = g.n2 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (g.valid) {
        const float4* const tr = t_rec + 2 * (size_t)g.j;
        g.pf = out[i], g.n1 = s_cnrm[i], g.qf = tr[0], g.n2 = tr[1];
      }
      return g;
    };
    nnkey_t kq[AHEAD];
#pragma unroll
    for (int r = 0; r < AHEAD; ++r) {
      const int i = first + r * RED_THREADS + lane;
      kq[r] = i < i1 ? keys[i] : NNKEY_NONE;
    }
    Gathered cur = gather(kq[0], first + lane);
    for (int base = first; base < i1; base += RED_THREADS) {  // wave-uniform
      const Gathered nxt = gather(kq[1], base + RED_THREADS + lane);
#pragma unroll
      for (int r = 0; r + 1 < AHEAD; ++r) kq[r] = kq[r + 1];
      {
        const int i = base + AHEAD * RED_THREADS + lane;
        kq[AHEAD - 1] = i < i1 ? keys[i] : NNKEY_NONE;
      }
      double s00 = 0, s01 = 0, s02 = 0, s11 = 0, s12 = 0, s22 = 0;
      if (cur.valid) {
        const float4 n1 = cur.n1, n2 = cur.n2;
        const double n1x = n1.x, n1y = n1.y, n1z = n1.z, n2x = n2.x, n2y = n2.y, n2z = n2.z;
        const double k1 = 0.999 * fast_rcp(n1x * n1x + n1y * n1y + n1z * n1z);
        const double k2 = 0.999 * fast_rcp(n2x * n2x + n2y * n2y + n2z * n2z);
        const double m0 = R[0] * n1x + R[1] * n1y + R[2] * n1z;
        const double m1 = R[3] * n1x + R[4] * n1y + R[5] * n1z;
        const double m2 = R[6] * n1x + R[7] * n1y + R[8] * n1z;
        const double u0 = k1 * m0, u1 = k1 * m1, u2 = k1 * m2;
        const double v0 = k2 * n2x, v1 = k2 * n2y, v2 = k2 * n2z;
        s00 = (G[0] - u0 * m0) - v0 * n2x;
        s01 = (G[1] - u0 * m1) - v0 * n2y;
        s02 = (G[2] - u0 * m2) - v0 * n2z;
        s11 = (G[3] - u1 * m1) - v1 * n2y;
        s12 = (G[4] - u1 * m2) - v1 * n2z;
        s22 = (G[5] - u2 * m2) - v2 * n2z;
        if (__builtin_expect(__ballot(n1.w != 0.f || n2.w != 0.f) != 0, 0)) {
          if (n1.w != 0.f || n2.w != 0.f) {
            double c1[6], c2[6];
            if (n1.w != 0.f) load_cov6(s_cov + 6 * (size_t)cur.i, c1); else cov_of_regular_normal(n1, c1);
            if (n2.w != 0.f) load_cov6(t_cov + 6 * (size_t)cur.j, c2); else cov_of_regular_normal(n2, c2);
            sandwich(R, c1, c2, s00, s01, s02, s11, s12, s22);
          }
        }
      }
      accumulate(cur.valid, cur.pf, cur.qf, s00, s01, s02, s11, s12, s22);
      cur = nxt;
    }
  } else {
    for (int gbase = first; gbase < i1; gbase += AHEAD * RED_THREADS) {  // wave-uniform
      nnkey_t kahead[AHEAD];
#pragma unroll
      for (int r = 0; r < AHEAD; ++r) {
        const int i = gbase + r * RED_THREADS + lane;
        kahead[r] = i < i1 ? keys[i] : NNKEY_NONE;
      }
#pragma unroll
      for (int r = 0; r < AHEAD; ++r) {
        const int base = gbase + r * RED_THREADS;
        if (base >= i1) break;
        const int i = base + lane;
        const nnkey_t key = kahead[r];
        const float d2 = __uint_as_float((unsigned)(key >> 32));
        const bool valid = key != NNKEY_NONE && (double)d2 < corr_dist2;  // gicp.hpp:483
        float4 pf = make_float4(0.f, 0.f, 0.f, 0.f), qf = pf;
        double s00 = 0, s01 = 0, s02 = 0, s11 = 0, s12 = 0, s22 = 0;
        if (valid) {
          // the matched target's point and covariance sit in one 64-byte record: one cache line
          const float4* const tr = t_rec + 4 * (size_t)(unsigned)(key & 0xFFFFFFFFu);
          pf = out[i];
          qf = tr[0];
          double c1[6], c2[6];
          load_cov6(s_cov + 6 * (size_t)i, c1);
          load_cov6(reinterpret_cast<const double*>(tr + 1), c2);
          sandwich(R, c1, c2, s00, s01, s02, s11, s12, s22);
        }
        accumulate(valid, pf, qf, s00, s01, s02, s11, s12, s22);
      }
    }
  }

  // D[row][col] sits in lane (row & 3) * 16 + col, register row >> 2; each wave parks its D in
  // its own staging rows, then one thread per statistic adds the four waves in order
  const red_d4 acc = acc0 + acc1;
#pragma unroll
  for (int q = 0; q < 4; ++q) sa[q * 64 + lane] = acc[q];
  __syncthreads();
  for (int sidx = threadIdx.x; sidx < NSTAT; sidx += RED_THREADS) {
    auto D = [&](int w, int row, int col) {
      return stage[w][(row >> 2) * 64 + (row & 3) * 16 + col];
    };
    double s = 0;
#pragma unroll
    for (int w = 0; w < RED_THREADS / 64; ++w) {
      if (sidx < 60) {
        s += D(w, sidx / 6, sidx % 6);
      } else if (sidx < 72) {
        const int k = (sidx - 60) / 3, a = (sidx - 60) % 3;
        s += D(w, uu_slot(k, 3), 6 + a);
      } else if (sidx == 72) {
        s += (D(w, 10, 6) + D(w, 11, 7)) + D(w, 12, 8);
      } else {
        s += D(w, 9, 9);
      }
    }
    partials[((size_t)pair * RED_MAX_CHUNKS + bm.y) * NSTAT + sidx] = s;
  }
}

hipError_t launch_gicp_reduce(const CloudDev* clouds, const PairDev* pairs, const PairState* st,
                              const int4* bmap, int n_blocks, double corr_dist2, double* partials,
                              bool compact, hipStream_t stream) {
  if (n_blocks <= 0) return hipSuccess;
  if (compact)
    hipLaunchKernelGGL(gicp_reduce_kernel<true>, dim3(n_blocks), dim3(RED_THREADS), 0, stream, clouds,
                       pairs, st, bmap, corr_dist2, partials);
  else
    hipLaunchKernelGGL(gicp_reduce_kernel<false>, dim3(n_blocks), dim3(RED_THREADS), 0, stream, clouds,
                       pairs, st, bmap, corr_dist2, partials);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Gauss-Newton on the quadratic form.  Theta index: 3*k + a, k = column (0..3), a = row.

__device__ __forceinline__ double lane_value(double v, int l) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Called by all 64 lanes of a wave with the same x.  The three angles go to three lanes, so the
// wave pays for one sincos instead of three in a row.
__device__ void theta_of_x(const double x[6], double Th[12], double dTh[3][9], bool want_d) {
  const int l3 = (int)(threadIdx.x & 63) % 3;
  double sv, cv;
  sincos(l3 == 0 ? x[3] : (l3 == 1 ? x[4] : x[5]), &sv, &cv);
  const double sf = lane_value(sv, 0), cf = lane_value(cv, 0);
  const double st = lane_value(sv, 1), ct = lane_value(cv, 1);
  const double sp = lane_value(sv, 2), cp = lane_value(cv, 2);
  // R = Rz(psi) Ry(theta) Rx(phi), row-major
  const double R[9] = {cp * ct, cp * st * sf - sp * cf, cp * st * cf + sp * sf,
                       sp * ct, sp * st * sf + cp * cf, sp * st * cf - cp * sf,
                       -st,     ct * sf,                ct * cf};
  for (int k = 0; k < 3; ++k)
    for (int a = 0; a < 3; ++a) Th[k * 3 + a] = R[a * 3 + k];
  Th[9] = x[0];
  Th[10] = x[1];
  Th[11] = x[2];
  if (want_d) {
    const double dphi[9] = {0, cp * st * cf + sp * sf, -cp * st * sf + sp * cf,
                            0, sp * st * cf - cp * sf, -sp * st * sf - cp * cf,
                            0, ct * cf,                -ct * sf};
    const double dth[9] = {-cp * st, cp * ct * sf, cp * ct * cf, -sp * st, sp * ct * sf,
                           sp * ct * cf, -ct,      -st * sf,     -st * cf};
    const double dpsi[9] = {-sp * ct, -sp * st * sf - cp * cf, -sp * st * cf + cp * sf,
                            cp * ct,  cp * st * sf - sp * cf,  cp * st * cf + sp * sf,
                            0,        0,                       0};
    for (int k = 0; k < 3; ++k)
      for (int a = 0; a < 3; ++a) {
        dTh[0][k * 3 + a] = dphi[a * 3 + k];
        dTh[1][k * 3 + a] = dth[a * 3 + k];
        dTh[2][k * 3 + a] = dpsi[a * 3 + k];
      }
  }
}

// Solves H dx = g by Cholesky (H symmetric 6x6, lower triangle read).  Every loop has constant
// bounds and is unrolled, so L lives in registers (run-time indexing would put it in scratch
// memory, and this code runs on the latency-critical path of every outer iteration).
__device__ __forceinline__ bool chol6_solve(const double* __restrict__ H, const double* __restrict__ g,
                                            double dx[6]) {
  double L[21];  // row-major lower triangle: (i, j) at i (i + 1) / 2 + j
  double inv[6];  // 1 / L(j, j): one division per column instead of one per entry
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = H[i * 6 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
      if (i == j) {
        if (!(s > 0)) return false;
        const double d = sqrt(s);
        L[i * (i + 1) / 2 + i] = d;
        inv[i] = 1.0 / d;
      } else {
        L[i * (i + 1) / 2 + j] = s * inv[j];
      }
    }
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double s = g[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= L[i * (i + 1) / 2 + k] * y[k];
    y[i] = s * inv[i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) s -= L[k * (k + 1) / 2 + i] * dx[k];
    dx[i] = s * inv[i];
  }
  return true;
}

// 64-bit hash of the 16 float words of a transformation (multiply-xorshift per word)
__device__ unsigned long long hash_T(const float T[16]) {
  unsigned long long h = 1469598103934665603ull;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    h ^= (unsigned long long)__float_as_uint(T[i]);
    h *= 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
  }
  return h;
}

// One wave per pair.  Every lane carries the same scalar state (x, f, Theta ...) and executes
// the same control flow; the dense pieces of a Gauss-Newton step (Q d, Q J, J^T Q J, J^T g) are
// spread over the lanes one output entry per lane, each entry summed in the same order a single
// thread would use, and exchanged through LDS.
struct SolveLds {
  double stat[NSTAT + 2];
  double Q[144];   // full symmetric 12 x 12
  double b[12];
  double s[12];    // Q d
  double dR[27];   // derivatives of the 9 rotation entries of Theta wrt the three angles
  double QJ[27];   // (Q dR^T): rows = rotation entries, columns = angles
  double H[36];
  double g[6];
};

// f(Theta) = c + d^T (2 b + Q d), d = Theta - Theta0; optionally the gradient wrt Theta
__device__ double quad_eval_wave(SolveLds& L, const double Theta0[12], double c, const double Th[12],
                                 double* grad12, int lane) {
  double d[12];
  for (int i = 0; i < 12; ++i) d[i] = Th[i] - Theta0[i];
  __syncthreads();  // L.s may still be read from the previous evaluation
  if (lane < 12) {
    double acc = 0;
    for (int j = 0; j < 12; ++j) acc += L.Q[lane * 12 + j] * d[j];
    L.s[lane] = acc;
  }
  __syncthreads();
  double f = c;
  for (int i = 0; i < 12; ++i) {
    const double si = L.s[i];
    f += d[i] * (2.0 * L.b[i] + si);
    if (grad12) grad12[i] = 2.0 * (L.b[i] + si);
  }
  return f;
}

__global__ __launch_bounds__(64) void gicp_solve_kernel(const PairDev* __restrict__ pairs,
                                                        PairState* __restrict__ st,
                                                        const int* __restrict__ n_chunks,
                                                        const double* __restrict__ partials,
                                                        SolveParams sp, int* __restrict__ n_done,
                                                        int iter, int* __restrict__ ticket) {
  __shared__ SolveLds L;
  const int pair = blockIdx.x;
  PairState& S = st[pair];
  const int lane = threadIdx.x;
  if (!S.active) {
    if (lane == 0) iteration_tick(ticket, (int)gridDim.x, n_done, iter);
    return;
  }
  // history of the cycle test: one entry per lane, fetched while the slabs are being summed
  const unsigned long long my_hist = S.Thash[lane & 15];
  {
    // every slab value is requested before the first one is used (a loop with a run-time trip
    // count waits for each load in turn: 16 memory latencies on the critical path of the iteration)
    const int nc = n_chunks[pair];
    const double* const base = partials + (size_t)pair * RED_MAX_CHUNKS * NSTAT;
    double v0[RED_MAX_CHUNKS], v1[RED_MAX_CHUNKS];
    const bool second = lane + 64 < NSTAT;
#pragma unroll
    for (int c = 0; c < RED_MAX_CHUNKS; ++c) {
      v0[c] = c < nc ? base[c * NSTAT + lane] : 0.0;
      v1[c] = (c < nc && second) ? base[c * NSTAT + lane + 64] : 0.0;
    }
    double a0 = 0, a1 = 0;
#pragma unroll
    for (int c = 0; c < RED_MAX_CHUNKS; ++c) {
      a0 = c < nc ? a0 + v0[c] : a0;
      a1 = c < nc ? a1 + v1[c] : a1;
    }
    L.stat[lane] = a0;
    if (second) L.stat[lane + 64] = a1;
  }
  __syncthreads();

  const PairDev& P = pairs[pair];
  const int m = (int)(L.stat[NSTAT - 1] + 0.5);
  float T[16], Tprev[16];
  for (int i = 0; i < 16; ++i) T[i] = S.T[i];
  for (int i = 0; i < 16; ++i) Tprev[i] = T[i];  // previous_transformation_ = transformation_
  bool done = false;
  bool failed = false, converged = false, cycle = false;
  double delta = 0.;
  int it_out = S.iterations;
  unsigned long long h_prev = 0;
  if (m < 4) {
    // NotEnoughPointsException -> break out of the loop, converged_ stays false
    // (gicp.hpp:225-233, 542-547)
    failed = true;
    done = true;
  } else {
    // expand the 10 x 6 packed moments to the full symmetric 12 x 12
    for (int e = lane; e < 144; e += 64) {
      const int i = e / 12, j = e % 12;
      const int k = i / 3, a = i % 3, l = j / 3, bb = j % 3;
      const int kk = k <= l ? k : l, ll = k <= l ? l : k;
      const int aa = k <= l ? a : bb, b2 = k <= l ? bb : a;
      const double* s6 = &L.stat[uu_slot(kk, ll) * 6];
      const int lo = aa < b2 ? aa : b2, hi = aa < b2 ? b2 : aa;
      // symmetric 3x3 stored as c00 c01 c02 c11 c12 c22
      const int idx = lo == 0 ? hi : (lo == 1 ? 2 + hi : 5);
      L.Q[e] = s6[idx];
    }
    if (lane < 12) L.b[lane] = L.stat[60 + lane];
    __syncthreads();
    const double qc = L.stat[72];
    double Theta0[12];
    for (int k = 0; k < 3; ++k)
      for (int a = 0; a < 3; ++a) Theta0[k * 3 + a] = (double)T[a * 4 + k];
    for (int a = 0; a < 3; ++a) Theta0[9 + a] = (double)T[a * 4 + 3];

    double x[6];
    x[0] = T[3];
    x[1] = T[7];
    x[2] = T[11];
    {
      // the two atan2 in two lanes, one call
      const bool odd = lane & 1;
      const double at = atan2(odd ? (double)T[4] : (double)T[9], odd ? (double)T[0] : (double)T[10]);
      x[3] = lane_value(at, 0);
      x[4] = asin(-(double)T[8]);
      x[5] = lane_value(at, 1);
    }
    double Th[12], dTh[3][9], g12[12];
    theta_of_x(x, Th, dTh, true);
    double f = quad_eval_wave(L, Theta0, qc, Th, g12, lane);
    for (int it = 0; it < sp.max_inner_iterations; ++it) {
      // H = 2 J^T Q J and g = -J^T grad with J = dTheta/dx: the translation columns of J are unit
      // vectors and its angle columns only touch the nine rotation entries, so the products
      // against those columns reduce to entries of Q and to nine-term sums.  Every sum runs over
      // the same terms in the same order as the dense 12-term version (whose other terms are
      // exact zeros).
      __syncthreads();
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int e = 0; e < 9; ++e) L.dR[c * 9 + e] = dTh[c][e];
        L.g[0] = -g12[9], L.g[1] = -g12[10], L.g[2] = -g12[11];
      }
      __syncthreads();
      if (lane < 27) {  // QJ[e][c] = sum_e2 Q[e][e2] dR[c][e2]
        const int e = lane / 3, c = lane % 3;
        double acc = 0;
#pragma unroll
        for (int e2 = 0; e2 < 9; ++e2) acc += L.Q[e * 12 + e2] * L.dR[c * 9 + e2];
        L.QJ[lane] = acc;
      } else if (lane < 36) {  // H[3 + i][j] = 2 sum_e dR[i][e] Q[e][9 + j]
        const int i = (lane - 27) / 3, j = (lane - 27) % 3;
        double acc = 0;
#pragma unroll
        for (int e = 0; e < 9; ++e) acc += L.dR[i * 9 + e] * L.Q[e * 12 + 9 + j];
        L.H[(3 + i) * 6 + j] = 2.0 * acc;
      } else if (lane < 39) {  // g[3 + i] = -sum_e dR[i][e] grad[e]
        const int i = lane - 36;
        double acc = 0;
#pragma unroll
        for (int e = 0; e < 9; ++e) acc += L.dR[i * 9 + e] * g12[e];
        L.g[3 + i] = -acc;
      } else if (lane < 48) {  // H[i][j] = 2 Q[9 + i][9 + j]
        const int i = (lane - 39) / 3, j = (lane - 39) % 3;
        L.H[i * 6 + j] = 2.0 * (L.Q[(9 + i) * 12 + 9 + j] * 1.0);
      }
      __syncthreads();
      if (lane < 9) {  // H[3 + i][3 + j] = 2 sum_e dR[i][e] QJ[e][j]
        const int i = lane / 3, j = lane % 3;
        double acc = 0;
#pragma unroll
        for (int e = 0; e < 9; ++e) acc += L.dR[i * 9 + e] * L.QJ[e * 3 + j];
        L.H[(3 + i) * 6 + 3 + j] = 2.0 * acc;
      }
      __syncthreads();
      double dx[6];
      if (!chol6_solve(L.H, L.g, dx)) break;
      double step = 1.0, fn = f, xn[6], gn[12];
      bool ok = false;
      for (int h = 0; h < 30; ++h) {
        double reach = 0;
        for (int c = 0; c < 6; ++c) {
          xn[c] = x[c] + step * dx[c];
          reach = fmax(reach, fabs(step * dx[c]));
        }
        // a trial closer to x than the convergence threshold below cannot change the outcome by
        // more than that threshold; without this exit a solve that has already converged pays
        // for all 30 halvings (each a full evaluation) before it gives up
        if (reach < 1e-12) break;
        // value, gradient and derivatives at the trial point in one go: an accepted point (the
        // full step, almost always) is the next iteration's starting point as it is
        theta_of_x(xn, Th, dTh, true);
        fn = quad_eval_wave(L, Theta0, qc, Th, gn, lane);
        if (fn < f) {
          ok = true;
          break;
        }
        step *= 0.5;
      }
      if (!ok) break;
      double mx = 0;
      for (int c = 0; c < 6; ++c) {
        mx = fmax(mx, fabs(xn[c] - x[c]));
        x[c] = xn[c];
      }
      f = fn;
      for (int i = 0; i < 12; ++i) g12[i] = gn[i];
      if (mx < 1e-9) break;
    }
    apply_state(x, T);
    // convergence test (gicp.hpp:526-541, 564-580)
    for (int k = 0; k < 4; ++k)
      for (int l = 0; l < 4; ++l) {
        const double ratio = (k < 3 && l < 3) ? 1. / sp.rotation_epsilon : 1. / sp.tf_epsilon;
        const double c_delta = ratio * fabs((double)(Tprev[k * 4 + l] - T[k * 4 + l]));
        if (c_delta > delta) delta = c_delta;
      }
    it_out = S.iterations + 1;
    // With tf_epsilon = 1e-10 the test delta < 1 asks for a bit-identical float translation;
    // the exact minimiser can instead settle into a short limit cycle in which one
    // correspondence flips and T alternates between float-adjacent matrices (periods 2 and 6
    // observed).  A transformation already used for the lookups within the last 16 iterations
    // proves that no further progress is possible: stop as converged (the reference's BFGS
    // reports NoProgress in the same spot).
    {
      // both hashes in one pass: even lanes hash T, odd lanes Tprev
      float W[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) W[i] = (lane & 1) ? Tprev[i] : T[i];
      const unsigned long long hv = hash_T(W);
      const unsigned long long h =
          ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(hv >> 32), 0) << 32) |
          (unsigned)__builtin_amdgcn_readlane((int)(unsigned)hv, 0);
      h_prev = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(hv >> 32), 1) << 32) |
               (unsigned)__builtin_amdgcn_readlane((int)(unsigned)hv, 1);
      const int n_hist = it_out - 1 < 16 ? it_out - 1 : 16;
      cycle = sp.cycle_exit && __ballot(lane < n_hist && my_hist == h) != 0;
    }
    const bool by_rule = it_out >= sp.max_iterations || delta < 1;
    cycle = cycle && !by_rule;
    if (by_rule || cycle) {
      converged = true;
      for (int i = 0; i < 16; ++i) Tprev[i] = T[i];
      done = true;
    }
  }
  __syncthreads();  // every lane has read S before lane 0 rewrites it
  if (lane != 0) return;
  S.m = m;
  S.m_sum += (double)m;
  if (failed) {
    S.failed = 1;
    S.converged = 0;
  } else {
    S.delta = delta;
    S.iterations = it_out;
    for (int i = 0; i < 16; ++i) S.T[i] = T[i];
    S.Thash[(it_out - 1) % 16] = h_prev;
    if (converged) S.converged = 1;
    if (cycle) S.cycle = 1;
  }
  for (int i = 0; i < 16; ++i) S.Tprev[i] = Tprev[i];
  if (done) {
    float Tf[16], G[16];
    for (int i = 0; i < 16; ++i) G[i] = P.guess[i];
    mat4f_mul(Tprev, G, Tf);  // final_transformation_ = previous_ * guess (gicp.hpp:583)
    for (int i = 0; i < 16; ++i) S.Tfinal[i] = Tf[i];
    S.active = 0;
    // per-pair "left the loop" flag: the iteration it happened in.  n_done lives in pinned host
    // memory; entry 0 receives the iteration number once every workgroup has got this far
    __hip_atomic_store(&n_done[1 + pair], iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  iteration_tick(ticket, (int)gridDim.x, n_done, iter);
}

hipError_t launch_gicp_solve(const PairDev* pairs, PairState* st, int n_pairs,
                             const int* n_chunks, const double* partials, SolveParams sp,
                             int* n_done, int iter, int* ticket, hipStream_t stream) {
  if (n_pairs <= 0) return hipSuccess;
  hipLaunchKernelGGL(gicp_solve_kernel, dim3(n_pairs), dim3(64), 0, stream, pairs, st, n_chunks,
                     partials, sp, n_done, iter, ticket);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// After the alignment: fitness = mean squared 1-NN distance of final*source in the target
// (pcl getFitnessScore), finiteness of the aligned points (IcpLoopComputation.cc:558-559) and
// Ap = sum H^T H, H = [(a x R n)^T, (R n)^T] (PointCloudUtils.cc:115-163) with a = normalised
// un-accumulated source point (PointCloudUtils.cc:39-59), n = un-accumulated target normal of
// its correspondence (zero when the index lies beyond that scan, PointCloudUtils.cc:138-146).
//
// POINT2POINT mode instead accumulates H = sum J^T J with the closed-form 3x6 Jacobian of
// ComputeICPCovariancePointPoint (loop_closure/src/IcpLoopComputation.cc:770-882) over the
// aligned points; roll/pitch/yaw are the ZYX Euler angles of the final rotation.
template <int POINT2POINT>
__global__ __launch_bounds__(POST_THREADS) void post_reduce_kernel(
    const CloudDev* __restrict__ clouds, const PairDev* __restrict__ pairs,
    const PairState* __restrict__ st, const int4* __restrict__ bmap,
    double* __restrict__ partials) {
  __shared__ double sh[POST_THREADS / 64][NPOST];
  const int4 bm = bmap[blockIdx.x];
  const int pair = bm.x;
  const PairDev& P = pairs[pair];
  const PairState& S = st[pair];
  const CloudDev& S1 = clouds[P.scan1_cloud];
  const CloudDev& S2 = clouds[P.scan2_cloud];
  const CloudDev& SRC = clouds[P.src_cloud];
  double R[9];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) R[a * 3 + b] = (double)S.Tfinal[a * 4 + b];
  const float tf[12] = {S.Tfinal[0], S.Tfinal[1], S.Tfinal[2],  S.Tfinal[3],
                        S.Tfinal[4], S.Tfinal[5], S.Tfinal[6],  S.Tfinal[7],
                        S.Tfinal[8], S.Tfinal[9], S.Tfinal[10], S.Tfinal[11]};
  const float fac = S1.factor;
  const float ox = -fac * S1.centroid[0], oy = -fac * S1.centroid[1], oz = -fac * S1.centroid[2];

  // gu::Rot3::Roll/Pitch/Yaw of the final rotation (POINT2POINT only)
  double cr = 1, sr = 0, cp = 1, sp = 0, cy = 1, sy = 0;
  if (POINT2POINT) {
    const double pitch = -asin(R[6]);
    double roll, yaw;
    if (cos(pitch) > 1e-6) {
      roll = atan2(R[7], R[8]);
      yaw = atan2(R[3], R[0]);
    } else {
      roll = atan2(-R[5], R[4]);
      yaw = 0.0;
    }
    sincos(roll, &sr, &cr);
    sincos(pitch, &sp, &cp);
    sincos(yaw, &sy, &cy);
  }

  double acc[NPOST];
#pragma unroll
  for (int i = 0; i < NPOST; ++i) acc[i] = 0.0;
  const int n = P.n_src;
  const int per = (n + bm.z - 1) / bm.z;
  const int i0 = bm.y * per, i1 = min(n, i0 + per);
  for (int i = i0 + threadIdx.x; i < i1; i += POST_THREADS) {
    const float4 p = SRC.xyz[i];
    const float ax = ((tf[0] * p.x + tf[1] * p.y) + tf[2] * p.z) + tf[3];
    const float ay = ((tf[4] * p.x + tf[5] * p.y) + tf[6] * p.z) + tf[7];
    const float az = ((tf[8] * p.x + tf[9] * p.y) + tf[10] * p.z) + tf[11];
    if (!(isfinite(ax) && isfinite(ay) && isfinite(az))) acc[2] = 1.0;
    const nnkey_t key = P.keys[i];
    int j = -1;
    if (key != NNKEY_NONE) {
      acc[0] += (double)__uint_as_float((unsigned)(key >> 32));
      acc[1] += 1.0;
      j = (int)(key & 0xFFFFFFFFu);
    }
    if (POINT2POINT) {
      const double p_x = ax, p_y = ay, p_z = az;
      const double t_x = tf[3], t_y = tf[7], t_z = tf[11];
      const double ex = (t_x - p_x + p_z * cp - p_x * cy * sp + p_y * sp * sy);
      const double ey = (p_y - t_y + p_x * (sr * sy - cp * cr * cy) +
                         p_y * (cy * sr + cp * cr * sy) - p_z * cr * sp);
      const double ez = (t_z - p_z + p_x * (cr * sy + cp * cy * sr) +
                         p_y * (cr * cy - cp * sr * sy) + p_z * sp * sr);
      double J[3][6];
      J[0][0] = 0.0;
      J[0][1] = -2.0 * (p_z * sp + p_x * cp * cy - p_y * cp * sy) * ex;
      J[0][2] = 2.0 * (p_y * cy * sp + p_x * sp * sy) * ex;
      J[0][3] = 2.0 * t_x - 2.0 * p_x + 2.0 * p_z * cp - 2.0 * p_x * cy * sp + 2.0 * p_y * sp * sy;
      J[0][4] = 0.0;
      J[0][5] = 0.0;
      J[1][0] = 2.0 * (p_x * (cr * sy + cp * cy * sr) + p_y * (cr * cy - cp * sr * sy) +
                       p_z * sp * sr) * ey;
      J[1][1] = -2.0 * (p_z * cp * cr - p_x * cr * cy * sp + p_y * cr * sp * sy) * ey;
      J[1][2] = 2.0 * (p_x * (cy * sr + cp * cr * sy) - p_y * (sr * sy - cp * cr * cy)) * ey;
      J[1][3] = 0.0;
      J[1][4] = 2.0 * t_y - 2.0 * p_y - 2.0 * p_x * (sr * sy - cp * cr * cy) -
                2.0 * p_y * (cy * sr + cp * cr * sy) + 2.0 * p_z * cr * sp;
      J[1][5] = 0.0;
      J[2][0] = -2.0 * (p_x * (sr * sy - cp * cr * cy) + p_y * (cy * sr + cp * cr * sy) -
                        p_z * cr * sp) * ez;
      J[2][1] = 2.0 * (p_z * cp * sr - p_x * cy * sp * sr + p_y * sp * sr * sy) * ez;
      J[2][2] = 2.0 * (p_x * (cr * cy - cp * sr * sy) - p_y * (cr * sy + cp * cy * sr)) * ez;
      J[2][3] = 0.0;
      J[2][4] = 0.0;
      J[2][5] = 2.0 * t_z - 2.0 * p_z + 2.0 * p_x * (cr * sy + cp * cy * sr) +
                2.0 * p_y * (cr * cy - cp * sr * sy) + 2.0 * p_z * sp * sr;
      int s_ = 3;
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c)
          acc[s_++] += J[0][r] * J[0][c] + J[1][r] * J[1][c] + J[2][r] * J[2][c];
    } else if (i < S1.n && j >= 0) {
      // normalised query point: pcl::transformPointCloud with diag(factor) and -factor*centroid
      const float4 s = S1.xyz[i];
      const double a0 = (double)(((fac * s.x + 0.f * s.y) + 0.f * s.z) + ox);
      const double a1 = (double)(((0.f * s.x + fac * s.y) + 0.f * s.z) + oy);
      const double a2 = (double)(((0.f * s.x + 0.f * s.y) + fac * s.z) + oz);
      double n0 = 0, n1 = 0, n2 = 0;
      if (j < S2.n) {
        const float4 nn = S2.nrm_x[j];
        n0 = nn.x;
        n1 = nn.y;
        n2 = nn.z;
      }
      const bool bad = isnan(a0) || isnan(a1) || isnan(a2) || isnan(n0) || isnan(n1) || isnan(n2);
      if (!bad) {
        const double r0 = R[0] * n0 + R[1] * n1 + R[2] * n2;
        const double r1 = R[3] * n0 + R[4] * n1 + R[5] * n2;
        const double r2 = R[6] * n0 + R[7] * n1 + R[8] * n2;
        const double H[6] = {a1 * r2 - a2 * r1, a2 * r0 - a0 * r2, a0 * r1 - a1 * r0, r0, r1, r2};
        int s_ = 3;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = r; c < 6; ++c) acc[s_++] += H[r] * H[c];
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NPOST; ++i) {
    const double v = wave_sum(acc[i]);
    if (lane == 0) sh[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < NPOST) {
    double s = 0;
#pragma unroll
    for (int w = 0; w < POST_THREADS / 64; ++w) s += sh[w][threadIdx.x];
    partials[((size_t)pair * RED_MAX_CHUNKS + bm.y) * NPOST + threadIdx.x] = s;
  }
}

hipError_t launch_post_reduce(const CloudDev* clouds, const PairDev* pairs, const PairState* st,
                              const int4* bmap, int n_blocks, double* partials, bool point2point,
                              hipStream_t stream) {
  if (n_blocks <= 0) return hipSuccess;
  if (point2point)
    hipLaunchKernelGGL(post_reduce_kernel<1>, dim3(n_blocks), dim3(POST_THREADS), 0, stream, clouds,
                       pairs, st, bmap, partials);
  else
    hipLaunchKernelGGL(post_reduce_kernel<0>, dim3(n_blocks), dim3(POST_THREADS), 0, stream, clouds,
                       pairs, st, bmap, partials);
  return hipGetLastError();
}

}  // namespace lgicp
