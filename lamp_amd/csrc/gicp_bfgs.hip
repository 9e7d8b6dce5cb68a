// gicp_bfgs.hip — the reference's own inner solver on the device (lgicp_config::solver =
// LGICP_SOLVER_BFGS): pcl::BFGS as estimateRigidTransformationBFGS drives it
// (lamp_utils/include/lamp_utils/gicp.hpp:217-287) on the reference's objective, evaluated the way
// the reference evaluates it (gicp.hpp:290-402): T(x) built in float by applyState, T * p rounded to
// float per point, residual and Mahalanobis product in double, one pass over all correspondences
// per function evaluation.
//
// Why it exists beside the Gauss-Newton back end (gicp.hip): the reference ends an inner solve when
// the line search can no longer tell values of that float objective apart (NoProgress), or at
// |g| < 1e-2, or after 50 steps — none of which is the minimum Gauss-Newton goes to.  Where the
// outer loop runs to its fixed point (base-station block, 200 iterations) the two agree to the
// reference's own noise; where it is cut short (robot block, 5 iterations) they do not.  This back
// end follows the reference evaluation by evaluation instead.
//
// pcl::BFGS is a port of GSL's vector_bfgs2 (Fletcher's line search with cubic interpolation); it is
// restated here from the published algorithm.  One workgroup per pair runs the whole inner solve:
// every thread carries the same optimiser state and takes the same branches, a function evaluation
// is a strided pass of the workgroup over the pair's correspondence records (prepared once per
// outer iteration by bfgs_prepare_kernel: transformed source point, matched target point and the
// Mahalanobis matrix, 72 bytes) followed by a fixed-order reduction, so the sums — and with them
// every decision of the line search — are the same in all threads and from run to run.
#include "gicp_shared.h"

namespace lgicp {

constexpr int BF_THREADS = 512;
constexpr int BF_WAVES = BF_THREADS / 64;
constexpr int BF_NSUM = 13;  // f, M r (3), p (M r)^T (9)

// ---------------------------------------------------------------------------------------------
// Records of one outer iteration: for every source point of the chunk the transformed point
// (guess * source, what the reference keeps in `output`, gicp.hpp:440), the matched target point
// and M = (R C1 R^T + C2)^-1 (gicp.hpp:483-497); a point without a correspondence inside
// corr_dist gets an all-zero record, which adds exact zeros to every sum.  The number of
// correspondences of the chunk goes to partials[(pair, chunk)][0].
constexpr int BFP_THREADS = 64;  // one wave per slab (the count below is a wave sum)

template <bool COMPACT>
__global__ __launch_bounds__(BFP_THREADS) void bfgs_prepare_kernel(
    const CloudDev* __restrict__ clouds, const PairDev* __restrict__ pairs,
    const PairState* __restrict__ st, const int4* __restrict__ bmap, double corr_dist2,
    double* __restrict__ partials) {
  const int4 bm = bmap[blockIdx.x];
  const int pair = bm.x;
  const PairState& S = st[pair];
  if (!S.active) return;
  const PairDev& P = pairs[pair];
  const CloudDev& SC = clouds[P.src_cloud];
  const CloudDev& TC = clouds[P.tgt_cloud];
  // R = rotation of transformation_ * guess in double (gicp.hpp:450-460)
  double R[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += (double)S.T[i * 4 + k] * (double)P.guess[k * 4 + j];
      R[i * 3 + j] = s;
    }
  const int n = P.n_src;
  const int per = (n + bm.z - 1) / bm.z;
  const int i0 = bm.y * per, i1 = min(n, i0 + per);
  int count = 0;
  for (int i = i0 + (int)threadIdx.x; i < i1; i += BFP_THREADS) {
    const nnkey_t key = P.keys[i];
    const float d2 = __uint_as_float((unsigned)(key >> 32));
    const bool valid = key != NNKEY_NONE && (double)d2 < corr_dist2;  // gicp.hpp:483
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 b = make_float2(0.f, 0.f);
    double2 m0 = make_double2(0., 0.), m1 = m0, m2 = m0;
    if (valid) {
      const int j = (int)(key & 0xFFFFFFFFu);
      double c1[6], c2[6];
      float4 qf;
      if (COMPACT) {
        const float4 n1 = SC.cnrm[i];
        const float4* const tr = TC.rec + 2 * (size_t)j;
        qf = tr[0];
        const float4 n2 = tr[1];
        if (n1.w != 0.f) load_cov6(SC.cov + 6 * (size_t)i, c1); else cov_of_regular_normal(n1, c1);
        if (n2.w != 0.f) load_cov6(TC.cov + 6 * (size_t)j, c2); else cov_of_regular_normal(n2, c2);
      } else {
        const float4* const tr = TC.rec + 4 * (size_t)j;
        qf = tr[0];
        load_cov6(SC.cov + 6 * (size_t)i, c1);
        load_cov6(reinterpret_cast<const double*>(tr + 1), c2);
      }
      double s00, s01, s02, s11, s12, s22;
      sandwich(R, c1, c2, s00, s01, s02, s11, s12, s22);
      // M = S^-1 by cofactors (Eigen's 3x3 inverse, gicp.hpp:493)
      const double k00 = s11 * s22 - s12 * s12;
      const double k01 = s02 * s12 - s01 * s22;
      const double k02 = s01 * s12 - s02 * s11;
      const double det = s00 * k00 + s01 * k01 + s02 * k02;
      const double inv = 1.0 / det;
      m0 = make_double2(k00 * inv, k01 * inv);
      m1 = make_double2(k02 * inv, (s00 * s22 - s02 * s02) * inv);
      m2 = make_double2((s01 * s02 - s00 * s12) * inv, (s00 * s11 - s01 * s01) * inv);
      const float4 pf = P.out[i];
      a = make_float4(pf.x, pf.y, pf.z, qf.x);
      b = make_float2(qf.y, qf.z);
      ++count;
    }
    P.bf_a[i] = a;
    P.bf_b[i] = b;
    double2* const mm = P.bf_m + 3 * (size_t)i;
    mm[0] = m0, mm[1] = m1, mm[2] = m2;
  }
  const double total = wave_sum((double)count);
  if (threadIdx.x == 0) partials[((size_t)pair * RED_MAX_CHUNKS + bm.y) * NSTAT] = total;
}

hipError_t launch_bfgs_prepare(const CloudDev* clouds, const PairDev* pairs, const PairState* st,
                               const int4* bmap, int n_blocks, double corr_dist2, double* partials,
                               bool compact, hipStream_t stream) {
  if (n_blocks <= 0) return hipSuccess;
  if (compact)
    hipLaunchKernelGGL(bfgs_prepare_kernel<true>, dim3(n_blocks), dim3(BFP_THREADS), 0, stream,
                       clouds, pairs, st, bmap, corr_dist2, partials);
  else
    hipLaunchKernelGGL(bfgs_prepare_kernel<false>, dim3(n_blocks), dim3(BFP_THREADS), 0, stream,
                       clouds, pairs, st, bmap, corr_dist2, partials);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// OptimizationFunctorWithIndices (gicp.hpp:290-402) over the records of one pair;
// base_transformation_ is the identity throughout (gicp.hpp:435).
struct RefObjective {
  const float4* a;
  const float2* b;
  const double2* mm;
  int n;
  double m;                      // number of correspondences
  double (*lds)[BF_WAVES][BF_NSUM];  // [2]: the buffers of successive evaluations take turns
  int turn;

  // computeRDerivative (gicp.hpp:159-214): g[3..5] = <dR/dangle, R>
  __device__ static void r_derivative(const double x[6], const double R[9], double g[6]) {
    const double phi = x[3], theta = x[4], psi = x[5];
    const double cphi = cos(phi), sphi = sin(phi);
    const double ctheta = cos(theta), stheta = sin(theta);
    const double cpsi = cos(psi), spsi = sin(psi);
    double dphi[9], dth[9], dpsi[9];
    dphi[0] = 0., dphi[3] = 0., dphi[6] = 0.;
    dphi[1] = sphi * spsi + cphi * cpsi * stheta;
    dphi[4] = -cpsi * sphi + cphi * spsi * stheta;
    dphi[7] = cphi * ctheta;
    dphi[2] = cphi * spsi - cpsi * sphi * stheta;
    dphi[5] = -cphi * cpsi - sphi * spsi * stheta;
    dphi[8] = -ctheta * sphi;
    dth[0] = -cpsi * stheta;
    dth[3] = -spsi * stheta;
    dth[6] = -ctheta;
    dth[1] = cpsi * ctheta * sphi;
    dth[4] = ctheta * sphi * spsi;
    dth[7] = -sphi * stheta;
    dth[2] = cphi * cpsi * ctheta;
    dth[5] = cphi * ctheta * spsi;
    dth[8] = -cphi * stheta;
    dpsi[0] = -ctheta * spsi;
    dpsi[3] = cpsi * ctheta;
    dpsi[6] = 0.;
    dpsi[1] = -cphi * cpsi - sphi * spsi * stheta;
    dpsi[4] = -cphi * spsi + cpsi * sphi * stheta;
    dpsi[7] = 0.;
    dpsi[2] = cpsi * sphi - cphi * spsi * stheta;
    dpsi[5] = sphi * spsi + cphi * cpsi * stheta;
    dpsi[8] = 0.;
    // matricesInnerProd (gicp.h:365-374): sum_ij mat1(j, i) * mat2(i, j)
    double g3 = 0, g4 = 0, g5 = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        g3 += dphi[j * 3 + i] * R[i * 3 + j];
        g4 += dth[j * 3 + i] * R[i * 3 + j];
        g5 += dpsi[j * 3 + i] * R[i * 3 + j];
      }
    g[3] = g3, g[4] = g4, g[5] = g5;
  }

  // value (WANT_G: and gradient) at x; every thread of the workgroup calls it with the same x
  // and receives the same numbers
  template <bool WANT_G>
  __device__ __attribute__((noinline)) void eval(const double x[6], double* f_out, double* g) {
    float T[16];
    apply_state<true>(x, T);
    constexpr int NS = WANT_G ? BF_NSUM : 1;
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
    constexpr int UNROLL = 4;
    for (int base = (int)threadIdx.x; base < n; base += UNROLL * BF_THREADS) {
      float4 ra[UNROLL];
      float2 rb[UNROLL];
      double2 r0[UNROLL], r1[UNROLL], r2[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int i = min(base + u * BF_THREADS, n - 1);
        ra[u] = a[i];
        rb[u] = b[i];
        const double2* const q = mm + 3 * (size_t)i;
        r0[u] = q[0], r1[u] = q[1], r2[u] = q[2];
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        if (base + u * BF_THREADS >= n) break;
        const float px = ra[u].x, py = ra[u].y, pz = ra[u].z;
        // Eigen::Vector4f pp(transformation_matrix * p_src), float (gicp.hpp:307)
        const float ppx = ((T[0] * px + T[1] * py) + T[2] * pz) + T[3] * 1.0f;
        const float ppy = ((T[4] * px + T[5] * py) + T[6] * pz) + T[7] * 1.0f;
        const float ppz = ((T[8] * px + T[9] * py) + T[10] * pz) + T[11] * 1.0f;
        const double e0 = (double)(ppx - ra[u].w), e1 = (double)(ppy - rb[u].x),
                     e2 = (double)(ppz - rb[u].y);
        const double M00 = r0[u].x, M01 = r0[u].y, M02 = r1[u].x, M11 = r1[u].y, M12 = r2[u].x,
                     M22 = r2[u].y;
        const double t0 = M00 * e0 + M01 * e1 + M02 * e2;
        const double t1 = M01 * e0 + M11 * e1 + M12 * e2;
        const double t2 = M02 * e0 + M12 * e1 + M22 * e2;
        acc[0] += e0 * t0 + e1 * t1 + e2 * t2;
        if (WANT_G) {
          acc[1] += t0, acc[2] += t1, acc[3] += t2;
          const double dx = (double)px, dy = (double)py, dz = (double)pz;
          acc[4] += dx * t0, acc[5] += dx * t1, acc[6] += dx * t2;
          acc[7] += dy * t0, acc[8] += dy * t1, acc[9] += dy * t2;
          acc[10] += dz * t0, acc[11] += dz * t1, acc[12] += dz * t2;
        }
      }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double (*buf)[BF_NSUM] = lds[turn];
    turn ^= 1;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const double v = wave_sum(acc[k]);
      if (lane == 0) buf[wave][k] = v;
    }
    __syncthreads();
    double tot[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      double s = 0;
#pragma unroll
      for (int w = 0; w < BF_WAVES; ++w) s += buf[w][k];
      tot[k] = s;
    }
    if (f_out) *f_out = tot[0] / m;
    if (WANT_G) {
      double Racc[9];
#pragma unroll
      for (int k = 0; k < 3; ++k) g[k] = tot[1 + k] * (2.0 / m);
#pragma unroll
      for (int k = 0; k < 9; ++k) Racc[k] = tot[4 + k] * (2.0 / m);
      r_derivative(x, Racc, g);
    }
  }
};

// ---------------------------------------------------------------------------------------------
// pcl::BFGS = GSL vector_bfgs2: the optimiser state, the line function with its caches (a value or
// a slope asked for twice at the same step length is evaluated once) and Fletcher's line search.
enum { BF_RUNNING = 0, BF_SUCCESS = 1, BF_NOPROGRESS = 2 };

__device__ __forceinline__ double dot6(const double* a, const double* b) {
  double s = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) s += a[i] * b[i];
  return s;
}
__device__ __forceinline__ double nrm6(const double* a) { return sqrt(dot6(a, a)); }
__device__ __forceinline__ void copy6(double* d, const double* s) {
#pragma unroll
  for (int i = 0; i < 6; ++i) d[i] = s[i];
}

template <class F>
struct Bfgs {
  F& fn;
  double x0[6], g0[6], p[6];
  double f, g[6];  // current value / gradient
  double g0norm, pnorm, delta_f, fp0, step;
  double x_alpha[6], g_alpha[6], f_alpha, df_alpha;
  double f_cache_key, df_cache_key, x_cache_key, g_cache_key;
  double rho, sigma, tau1, tau2, tau3;

  __device__ explicit Bfgs(F& f_) : fn(f_) {}

  __device__ void move_to(double alpha) {
    if (alpha == x_cache_key) return;
#pragma unroll
    for (int i = 0; i < 6; ++i) x_alpha[i] = x0[i] + alpha * p[i];
    x_cache_key = alpha;
  }
  __device__ double slope() { return dot6(g_alpha, p); }
  __device__ double line_f(double alpha) {
    if (alpha == f_cache_key) return f_alpha;
    move_to(alpha);
    fn.template eval<false>(x_alpha, &f_alpha, nullptr);
    f_cache_key = alpha;
    return f_alpha;
  }
  __device__ double line_df(double alpha) {
    if (alpha == df_cache_key) return df_alpha;
    move_to(alpha);
    if (alpha != g_cache_key) {
      fn.template eval<true>(x_alpha, nullptr, g_alpha);
      g_cache_key = alpha;
    }
    df_alpha = slope();
    df_cache_key = alpha;
    return df_alpha;
  }
  __device__ void line_fdf(double alpha, double* fo, double* dfo) {
    if (alpha == f_cache_key && alpha == df_cache_key) {
      *fo = f_alpha;
      *dfo = df_alpha;
      return;
    }
    if (alpha == f_cache_key || alpha == df_cache_key) {
      *fo = line_f(alpha);
      *dfo = line_df(alpha);
      return;
    }
    move_to(alpha);
    fn.template eval<true>(x_alpha, &f_alpha, g_alpha);
    f_cache_key = alpha;
    g_cache_key = alpha;
    df_alpha = slope();
    df_cache_key = alpha;
    *fo = f_alpha;
    *dfo = df_alpha;
  }

  __device__ static double cubic(double c0, double c1, double c2, double c3, double z) {
    return c0 + z * (c1 + z * (c2 + z * c3));
  }
  __device__ static void check_extremum(double c0, double c1, double c2, double c3, double z,
                                        double* zmin, double* fmin) {
    const double y = cubic(c0, c1, c2, c3, z);
    if (y < *fmin) {
      *zmin = z;
      *fmin = y;
    }
  }
  // gsl_poly_solve_quadratic: number of real roots, ascending
  __device__ static int solve_quadratic(double qa, double qb, double qc, double* r0, double* r1) {
    if (qa == 0) {
      if (qb == 0) return 0;
      *r0 = -qc / qb;
      return 1;
    }
    const double disc = qb * qb - 4 * qa * qc;
    if (disc > 0) {
      if (qb == 0) {
        const double r = sqrt(-qc / qa);
        *r0 = -r;
        *r1 = r;
      } else {
        const double sgnb = (qb > 0 ? 1 : -1);
        const double temp = -0.5 * (qb + sgnb * sqrt(disc));
        const double a1 = temp / qa, a2 = qc / temp;
        if (a1 < a2) {
          *r0 = a1;
          *r1 = a2;
        } else {
          *r0 = a2;
          *r1 = a1;
        }
      }
      return 2;
    } else if (disc == 0) {
      *r0 = -0.5 * qb / qa;
      *r1 = -0.5 * qb / qa;
      return 2;
    }
    return 0;
  }
  __device__ static double interp_quad(double f0, double fp, double f1, double zl, double zh) {
    const double fl = f0 + zl * (fp + zl * (f1 - f0 - fp));
    const double fh = f0 + zh * (fp + zh * (f1 - f0 - fp));
    const double c = 2 * (f1 - f0 - fp);
    double zmin = zl, fmin = fl;
    if (fh < fmin) {
      zmin = zh;
      fmin = fh;
    }
    if (c > 0) {
      const double z = -fp / c;
      if (z > zl && z < zh) {
        const double fz = f0 + z * (fp + z * (f1 - f0 - fp));
        if (fz < fmin) {
          zmin = z;
          fmin = fz;
        }
      }
    }
    return zmin;
  }
  __device__ static double interp_cubic(double f0, double fp, double f1, double fp1, double zl,
                                        double zh) {
    const double eta = 3 * (f1 - f0) - 2 * fp - fp1;
    const double xi = fp + fp1 - 2 * (f1 - f0);
    const double c0 = f0, c1 = fp, c2 = eta, c3 = xi;
    double zmin = zl, fmin = cubic(c0, c1, c2, c3, zl), z0 = 0, z1 = 0;
    check_extremum(c0, c1, c2, c3, zh, &zmin, &fmin);
    const int nr = solve_quadratic(3 * c3, 2 * c2, c1, &z0, &z1);
    if (nr == 2) {
      if (z0 > zl && z0 < zh) check_extremum(c0, c1, c2, c3, z0, &zmin, &fmin);
      if (z1 > zl && z1 < zh) check_extremum(c0, c1, c2, c3, z1, &zmin, &fmin);
    } else if (nr == 1) {
      if (z0 > zl && z0 < zh) check_extremum(c0, c1, c2, c3, z0, &zmin, &fmin);
    }
    return zmin;
  }
  // order 3 (gicp.hpp:258): cubic wherever the slope at b is known
  __device__ static double interpolate(double a, double fa, double fpa, double b, double fb,
                                       double fpb, double xmin, double xmax) {
    double zmin = (xmin - a) / (b - a), zmax = (xmax - a) / (b - a), z;
    if (zmin > zmax) {
      const double t = zmin;
      zmin = zmax;
      zmax = t;
    }
    if (isfinite(fpb))
      z = interp_cubic(fa, fpa * (b - a), fb, fpb * (b - a), zmin, zmax);
    else
      z = interp_quad(fa, fpa * (b - a), fb, zmin, zmax);
    return a + z * (b - a);
  }

  // pcl::BFGS::lineSearch (GSL minimize()): bracketing, then sectioning, strong Wolfe conditions
  __device__ int line_search(double alpha1, double* alpha_new) {
    double f0, fp0_, falpha, falpha_prev, fpalpha = 0, fpalpha_prev, delta, alpha_next;
    double alpha = alpha1, alpha_prev = 0.0;
    double a, b, fa, fb, fpa, fpb;
    const int bracket_iters = 100, section_iters = 100;
    int i = 0;
    line_fdf(0.0, &f0, &fp0_);
    falpha_prev = f0;
    fpalpha_prev = fp0_;
    a = 0.0, b = alpha;
    fa = f0, fb = 0.0;
    fpa = fp0_, fpb = 0.0;
    while (i++ < bracket_iters) {
      falpha = line_f(alpha);
      if (falpha > f0 + alpha * rho * fp0_ || falpha >= falpha_prev) {
        a = alpha_prev, fa = falpha_prev, fpa = fpalpha_prev;
        b = alpha, fb = falpha, fpb = __builtin_nan("");
        break;
      }
      fpalpha = line_df(alpha);
      if (fabs(fpalpha) <= -sigma * fp0_) {
        *alpha_new = alpha;
        return BF_SUCCESS;
      }
      if (fpalpha >= 0) {
        a = alpha, fa = falpha, fpa = fpalpha;
        b = alpha_prev, fb = falpha_prev, fpb = fpalpha_prev;
        break;
      }
      delta = alpha - alpha_prev;
      {
        const double lower = alpha + delta, upper = alpha + tau1 * delta;
        alpha_next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha,
                                 lower, upper);
      }
      alpha_prev = alpha;
      falpha_prev = falpha;
      fpalpha_prev = fpalpha;
      alpha = alpha_next;
    }
    while (i++ < section_iters) {
      delta = b - a;
      {
        const double lower = a + tau2 * delta, upper = b - tau3 * delta;
        alpha = interpolate(a, fa, fpa, b, fb, fpb, lower, upper);
      }
      falpha = line_f(alpha);
      if ((a - alpha) * fpa <= 2.2204460492503131e-16) return BF_NOPROGRESS;  // roundoff prevents progress
      if (falpha > f0 + rho * alpha * fp0_ || falpha >= fa) {
        b = alpha, fb = falpha, fpb = __builtin_nan("");
      } else {
        fpalpha = line_df(alpha);
        if (fabs(fpalpha) <= -sigma * fp0_) {
          *alpha_new = alpha;
          return BF_SUCCESS;
        }
        if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) {
          b = a, fb = fa, fpb = fpa;
          a = alpha, fa = falpha, fpa = fpalpha;
        } else {
          a = alpha, fa = falpha, fpa = fpalpha;
        }
      }
    }
    return BF_SUCCESS;
  }

  // minimizeInit
  __device__ void init(const double x[6]) {
    step = 1.0;
    rho = 0.01, sigma = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5;  // gicp.hpp:253-258
    delta_f = 0;
    fn.template eval<true>(x, &f, g);
    copy6(x0, x);
    copy6(g0, g);
    g0norm = nrm6(g0);
#pragma unroll
    for (int i = 0; i < 6; ++i) p[i] = g[i] * -1 / g0norm;
    pnorm = nrm6(p);
    fp0 = -g0norm;
    copy6(x_alpha, x0);
    copy6(g_alpha, g0);
    f_alpha = f;
    df_alpha = slope();
    f_cache_key = df_cache_key = x_cache_key = g_cache_key = 0.0;
  }

  // minimizeOneStep: BF_RUNNING = a step was taken
  __device__ int one_step(double x[6]) {
    double alpha = 0.0, alpha1;
    const double f0 = f;
    if (pnorm == 0.0 || g0norm == 0.0 || fp0 == 0 || !isfinite(fp0)) return BF_NOPROGRESS;
    if (delta_f < 0) {
      const double del = fmax(-delta_f, 10 * 2.2204460492503131e-16 * fabs(f0));
      alpha1 = fmin(1.0, 2.0 * del / (-fp0));
    } else {
      alpha1 = fabs(step);
    }
    const int status = line_search(alpha1, &alpha);
    if (status != BF_SUCCESS) return status;
    {
      double fa_, dfa_;
      line_fdf(alpha, &fa_, &dfa_);  // update_position
      f = fa_;
      copy6(x, x_alpha);
      copy6(g, g_alpha);
    }
    delta_f = f - f0;
    {
      double dx0[6], dg0[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        dx0[i] = x[i] - x0[i];
        dg0[i] = g[i] - g0[i];
      }
      const double dxg = dot6(dx0, g), dgg = dot6(dg0, g);
      const double dxdg = dot6(dx0, dg0), dgnorm = nrm6(dg0);
      double A, B;
      if (dxdg != 0) {
        B = dxg / dxdg;
        A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg;
      } else {
        B = 0;
        A = 0;
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) p[i] = g[i] - A * dx0[i] - B * dg0[i];
    }
    copy6(g0, g);
    copy6(x0, x);
    g0norm = nrm6(g0);
    pnorm = nrm6(p);
    const double pg = dot6(p, g);
    const double dir = (pg >= 0.0) ? -1.0 : +1.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) p[i] *= dir / pnorm;
    pnorm = nrm6(p);
    fp0 = dot6(p, g0);
    // change_direction: the caches now describe alpha = 0 of the new line
    copy6(x_alpha, x0);
    x_cache_key = 0.0;
    f_cache_key = 0.0;
    copy6(g_alpha, g0);
    g_cache_key = 0.0;
    df_alpha = slope();
    df_cache_key = 0.0;
    return BF_RUNNING;
  }
};

// One workgroup per pair: estimateRigidTransformationBFGS (gicp.hpp:217-287), applyState and the
// convergence test of the outer loop (gicp.hpp:526-541, 564-580).
__global__ __launch_bounds__(BF_THREADS) void gicp_bfgs_kernel(
    const PairDev* __restrict__ pairs, PairState* __restrict__ st, const int* __restrict__ n_chunks,
    const double* __restrict__ partials, SolveParams sp, int* __restrict__ n_done, int iter,
    int* __restrict__ ticket) {
  __shared__ double sums[2][BF_WAVES][BF_NSUM];
  const int pair = blockIdx.x;
  PairState& S = st[pair];
  if (!S.active) {
    if (threadIdx.x == 0) iteration_tick(ticket, (int)gridDim.x, n_done, iter);
    return;
  }
  const PairDev& P = pairs[pair];
  double mcount = 0;
  {
    const int nc = n_chunks[pair];
    for (int c = 0; c < nc; ++c) mcount += partials[((size_t)pair * RED_MAX_CHUNKS + c) * NSTAT];
  }
  const int m = (int)(mcount + 0.5);
  float T[16], Tprev[16];
  for (int i = 0; i < 16; ++i) T[i] = S.T[i];
  for (int i = 0; i < 16; ++i) Tprev[i] = T[i];  // previous_transformation_ = transformation_
  bool done = false, failed = false, converged = false;
  double delta = 0.;
  int it_out = S.iterations;
  if (m < 4) {
    // NotEnoughPointsException -> break out of the loop, converged_ stays false
    // (gicp.hpp:225-233, 542-547)
    failed = true;
    done = true;
  } else {
    RefObjective fn{P.bf_a, P.bf_b, P.bf_m, P.n_src, (double)m, sums, 0};
    double x[6];
    x[0] = T[3];
    x[1] = T[7];
    x[2] = T[11];
    x[3] = atan2((double)T[9], (double)T[10]);
    x[4] = asin(-(double)T[8]);
    x[5] = atan2((double)T[4], (double)T[0]);
    Bfgs<RefObjective> opt(fn);
    opt.init(x);
    int inner = 0, result;
    const double gradient_tol = 1e-2;  // gicp.hpp:250
    do {
      inner++;
      result = opt.one_step(x);
      if (result) break;
      result = (nrm6(opt.g) < gradient_tol) ? BF_SUCCESS : BF_RUNNING;  // testGradient
    } while (result == BF_RUNNING && inner < sp.max_inner_iterations);
    // NoProgress, Success and the step limit are all accepted (gicp.hpp:272-278)
    apply_state<true>(x, T);
    for (int k = 0; k < 4; ++k)
      for (int l = 0; l < 4; ++l) {
        const double ratio = (k < 3 && l < 3) ? 1. / sp.rotation_epsilon : 1. / sp.tf_epsilon;
        const double c_delta = ratio * fabs((double)(Tprev[k * 4 + l] - T[k * 4 + l]));
        if (c_delta > delta) delta = c_delta;
      }
    it_out = S.iterations + 1;
    if (it_out >= sp.max_iterations || delta < 1) {
      converged = true;
      for (int i = 0; i < 16; ++i) Tprev[i] = T[i];
      done = true;
    }
  }
  __syncthreads();  // every thread has read S before thread 0 rewrites it
  if (threadIdx.x != 0) return;
  S.m = m;
  S.m_sum += (double)m;
  if (failed) {
    S.failed = 1;
    S.converged = 0;
  } else {
    S.delta = delta;
    S.iterations = it_out;
    for (int i = 0; i < 16; ++i) S.T[i] = T[i];
    if (converged) S.converged = 1;
  }
  for (int i = 0; i < 16; ++i) S.Tprev[i] = Tprev[i];
  if (done) {
    float Tf[16], G[16];
    for (int i = 0; i < 16; ++i) G[i] = P.guess[i];
    mat4f_mul(Tprev, G, Tf);  // final_transformation_ = previous_ * guess (gicp.hpp:583)
    for (int i = 0; i < 16; ++i) S.Tfinal[i] = Tf[i];
    S.active = 0;
    // pinned host memory: the iteration in which the pair left the loop
    __hip_atomic_store(&n_done[1 + pair], iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  iteration_tick(ticket, (int)gridDim.x, n_done, iter);
}

hipError_t launch_gicp_bfgs(const PairDev* pairs, PairState* st, int n_pairs, const int* n_chunks,
                            const double* partials, SolveParams sp, int* n_done, int iter,
                            int* ticket, hipStream_t stream) {
  if (n_pairs <= 0) return hipSuccess;
  hipLaunchKernelGGL(gicp_bfgs_kernel, dim3(n_pairs), dim3(BF_THREADS), 0, stream, pairs, st,
                     n_chunks, partials, sp, n_done, iter, ticket);
  return hipGetLastError();
}

}  // namespace lgicp
