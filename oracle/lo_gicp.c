/*
 * lo_gicp.c — CPU restatement of LAMP's multithreaded Generalized-ICP
 * (TEST INFRASTRUCTURE ONLY, see lamp_oracle.h).
 *
 * Follows lamp_utils/include/lamp_utils/gicp.hpp, gicp.h and gicp_utils.h of the reference
 * line by line; the PCL base-class behaviour it relies on (Registration::align,
 * transformPointCloud, getFitnessScore, pcl::BFGS) is restated from PCL's published code,
 * pcl::BFGS being a port of GSL's vector_bfgs2 minimiser (Fletcher's line search).
 */
#include "lamp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ transforms */

/* pcl::transformPointCloud (PCL common/impl/transforms.hpp, dense branch), float matrix:
 * out.x = m00*x + m01*y + m02*z + m03 evaluated left to right
 * (call sites gicp.hpp:440,586). */
void lo_transform_points_f32(const float T[16], const float* xyz, int n, float* out) {
  for (int i = 0; i < n; ++i) {
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    out[3 * i + 0] = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
    out[3 * i + 1] = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
    out[3 * i + 2] = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
  }
}

/* same with Scalar = double (loop_closure/src/IcpLoopComputation.cc:926,944 pass a
 * gtsam Matrix4): computed in double, cast to float */
void lo_transform_points_f64(const double T[16], const float* xyz, int n, float* out) {
  for (int i = 0; i < n; ++i) {
    const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    out[3 * i + 0] = (float)(((T[0] * x + T[1] * y) + T[2] * z) + T[3]);
    out[3 * i + 1] = (float)(((T[4] * x + T[5] * y) + T[6] * z) + T[7]);
    out[3 * i + 2] = (float)(((T[8] * x + T[9] * y) + T[10] * z) + T[11]);
  }
}

/* Eigen::Matrix4f * Eigen::Matrix4f (gicp.hpp:583): coefficient (i,j) = sum_k a(i,k) b(k,j),
 * k ascending */
void lo_mat4f_mul(const float a[16], const float b[16], float out[16]) {
  float r[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float s = a[i * 4 + 0] * b[0 * 4 + j];
      s = s + a[i * 4 + 1] * b[1 * 4 + j];
      s = s + a[i * 4 + 2] * b[2 * 4 + j];
      s = s + a[i * 4 + 3] * b[3 * 4 + j];
      r[i * 4 + j] = s;
    }
  memcpy(out, r, sizeof(r));
}

/* Eigen::Matrix4f * Vector4f with w = 1 (gicp.hpp:469, :307): col0*x + col1*y + col2*z + col3 */
static inline void xform4(const float T[16], const float p[3], float q[3]) {
  q[0] = ((T[0] * p[0] + T[1] * p[1]) + T[2] * p[2]) + T[3] * 1.0f;
  q[1] = ((T[4] * p[0] + T[5] * p[1]) + T[6] * p[2]) + T[7] * 1.0f;
  q[2] = ((T[8] * p[0] + T[9] * p[1]) + T[10] * p[2]) + T[11] * 1.0f;
}

/* ------------------------------------------------------------------ covariances */

/* PCLTwoPlaneVectorsFromNormal<double> (gicp_utils.h:28-48) */
static void cov_from_normal(const float nrm[3], double C[9]) {
  double n[3] = {nrm[0], nrm[1], nrm[2]};
  double z = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
  if (z > 0) { /* Eigen >= 3.3 normalize(): no-op on a zero vector */
    const double s = sqrt(z);
    n[0] /= s;
    n[1] /= s;
    n[2] /= s;
  }
  double v1[3], v2[3];
  v1[0] = 1.0;
  v1[1] = 0.0;
  if (fabsf(nrm[2]) < 0.0000001 || fabs(n[2]) < 0.0000001) n[2] = 0.0000001;
  v1[2] = -n[0] / n[2];
  {
    const double s = sqrt(v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2]);
    v1[0] /= s;
    v1[1] /= s;
    v1[2] /= s;
  }
  v2[0] = n[1] * v1[2] - n[2] * v1[1];
  v2[1] = n[2] * v1[0] - n[0] * v1[2];
  v2[2] = n[0] * v1[1] - n[1] * v1[0];
  {
    const double z2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
    if (z2 > 0) {
      const double s = sqrt(z2);
      v2[0] /= s;
      v2[1] /= s;
      v2[2] /= s;
    }
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      C[i * 3 + j] = 0.001 * n[i] * n[j] + v1[i] * v1[j] + v2[i] * v2[j];
}

/* CalculateCovarianceFromNormals (gicp_utils.h:50-62) */
void lo_cov_from_normals(const float* normals, int n, double* cov9) {
  for (int i = 0; i < n; ++i) cov_from_normal(normals + 3 * i, cov9 + 9 * (size_t)i);
}

/* computeCovariances, k-NN branch (gicp.hpp:83-155) */
int lo_cov_knn(const float* xyz, int n, int k, double eps, double* cov9, int threads) {
  if (k > n) return -1; /* gicp.hpp:72-79 */
  lo_kdtree* tree = lo_kdtree_build(xyz, n);
  int32_t* nn = (int32_t*)malloc(sizeof(int32_t) * (size_t)n * k);
  float* nd = (float*)malloc(sizeof(float) * (size_t)n * k);
  lo_kdtree_knn(tree, xyz, n, k, nn, nd, threads);
  for (int i = 0; i < n; ++i) {
    double mean[3] = {0, 0, 0}, c[9] = {0};
    for (int j = 0; j < k; ++j) {
      const float* pt = xyz + 3 * nn[(size_t)i * k + j];
      /* pt.x * pt.x is a float product, accumulated into double (gicp.hpp:115-126) */
      mean[0] += pt[0];
      mean[1] += pt[1];
      mean[2] += pt[2];
      c[0] += pt[0] * pt[0];
      c[3] += pt[1] * pt[0];
      c[4] += pt[1] * pt[1];
      c[6] += pt[2] * pt[0];
      c[7] += pt[2] * pt[1];
      c[8] += pt[2] * pt[2];
    }
    for (int d = 0; d < 3; ++d) mean[d] /= (double)k;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b <= a; ++b) {
        c[a * 3 + b] /= (double)k;
        c[a * 3 + b] -= mean[a] * mean[b];
        c[b * 3 + a] = c[a * 3 + b];
      }
    /* JacobiSVD of a symmetric PSD matrix: U = eigenvectors, singular values descending */
    double ev[3], V[9];
    lo_sym_eig(c, 3, ev, V); /* ascending: column 0 is the smallest */
    double* C = cov9 + 9 * (size_t)i;
    for (int a = 0; a < 9; ++a) C[a] = 0.0;
    for (int col = 0; col < 3; ++col) {
      const double v = (col == 0) ? eps : 1.0; /* smallest singular value -> gicp_epsilon */
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) C[a * 3 + b] += v * V[a * 3 + col] * V[b * 3 + col];
    }
  }
  free(nn);
  free(nd);
  lo_kdtree_free(tree);
  return 0;
}

/* ------------------------------------------------------------------ optimisation functor */

typedef struct {
  const float* src; /* guess-transformed source ("output" cloud) */
  const float* tgt;
  const int32_t* si;
  const int32_t* ti;
  int m;
  const double* mahal; /* [n_src][9] */
  long n_evals;
  int residual_double; /* measurement knob: 1 = T*p in double instead of the reference's float */
  int sum_order;       /* measurement knob: 1 = the sums in the order of a 512-lane parallel sum */
} lo_functor;

/* applyState (gicp.hpp:619-634) on t = identity:
 * R = AngleAxisf(x5,Z) * AngleAxisf(x4,Y) * AngleAxisf(x3,X) — Eigen multiplies AngleAxis
 * objects as float quaternions and converts the product to a matrix. */
void lo_apply_state(const double x[6], float T[16]) {
  const float az = (float)x[5], ay = (float)x[4], ax = (float)x[3];
  /* quaternions (w, x, y, z) */
  const float qz[4] = {cosf(0.5f * az), 0.f, 0.f, sinf(0.5f * az)};
  const float qy[4] = {cosf(0.5f * ay), 0.f, sinf(0.5f * ay), 0.f};
  const float qx[4] = {cosf(0.5f * ax), sinf(0.5f * ax), 0.f, 0.f};
  float a[4], q[4];
#define QMUL(o, p, r)                                                  \
  do {                                                                 \
    o[0] = p[0] * r[0] - p[1] * r[1] - p[2] * r[2] - p[3] * r[3];      \
    o[1] = p[0] * r[1] + p[1] * r[0] + p[2] * r[3] - p[3] * r[2];      \
    o[2] = p[0] * r[2] + p[2] * r[0] + p[3] * r[1] - p[1] * r[3];      \
    o[3] = p[0] * r[3] + p[3] * r[0] + p[1] * r[2] - p[2] * r[1];      \
  } while (0)
  QMUL(a, qz, qy);
  QMUL(q, a, qx);
#undef QMUL
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
  /* t.col(3) += T; on identity */
  T[3] = 0.f + (float)x[0];
  T[7] = 0.f + (float)x[1];
  T[11] = 0.f + (float)x[2];
  T[15] = 1.f + 0.f;
}

/* computeRDerivative (gicp.hpp:159-214): g[3..5] = <dR/dangle, R> */
static void r_derivative(const double x[6], const double R[9], double g[6]) {
  const double phi = x[3], theta = x[4], psi = x[5];
  const double cphi = cos(phi), sphi = sin(phi);
  const double ctheta = cos(theta), stheta = sin(theta);
  const double cpsi = cos(psi), spsi = sin(psi);
  double dphi[9], dth[9], dpsi[9];
  dphi[0] = 0.;
  dphi[3] = 0.;
  dphi[6] = 0.;
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
  /* matricesInnerProd (gicp.h:365-374): sum_ij mat1(j,i) * mat2(i,j) */
  double g3 = 0, g4 = 0, g5 = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      g3 += dphi[j * 3 + i] * R[i * 3 + j];
      g4 += dth[j * 3 + i] * R[i * 3 + j];
      g5 += dpsi[j * 3 + i] * R[i * 3 + j];
    }
  g[3] = g3;
  g[4] = g4;
  g[5] = g5;
}

/* OptimizationFunctorWithIndices::fdf / operator() / df (gicp.hpp:290-402).
 * base_transformation_ is the identity throughout (gicp.hpp:435). */
/* per-correspondence terms: res^T M res, M res, p (M res)^T */
static inline void functor_terms(const lo_functor* F, const float T[16], int i, double t[13]) {
  const float* ps = F->src + 3 * F->si[i];
  const float* pt = F->tgt + 3 * F->ti[i];
  float pp[3];
  xform4(T, ps, pp);
  double res[3] = {(double)(pp[0] - pt[0]), (double)(pp[1] - pt[1]), (double)(pp[2] - pt[2])};
  if (F->residual_double)
    for (int a = 0; a < 3; ++a)
      res[a] = ((double)T[a * 4] * (double)ps[0] + (double)T[a * 4 + 1] * (double)ps[1] +
                (double)T[a * 4 + 2] * (double)ps[2] + (double)T[a * 4 + 3]) -
               (double)pt[a];
  const double* M = F->mahal + 9 * (size_t)F->si[i];
  const double temp[3] = {M[0] * res[0] + M[1] * res[1] + M[2] * res[2],
                          M[3] * res[0] + M[4] * res[1] + M[5] * res[2],
                          M[6] * res[0] + M[7] * res[1] + M[8] * res[2]};
  t[0] = res[0] * temp[0] + res[1] * temp[1] + res[2] * temp[2];
  t[1] = temp[0];
  t[2] = temp[1];
  t[3] = temp[2];
  /* pp = base_transformation_ * p_src = p_src ; R += p_src3 * temp^T */
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) t[4 + a * 3 + b] = (double)ps[a] * temp[b];
}

static void functor_fdf(lo_functor* F, const double x[6], double* f_out, double* g /*nullable*/) {
  float T[16];
  lo_apply_state(x, T);
  double tot[13] = {0};
  if (!F->sum_order) {
    /* the reference: one running sum per quantity, correspondences in order */
    for (int i = 0; i < F->m; ++i) {
      double t[13];
      functor_terms(F, T, i, t);
      tot[0] += t[0];
      if (g)
        for (int c = 1; c < 13; ++c) tot[c] += t[c];
    }
  } else {
    /* measurement knob: the same additions in the order of a 512-lane parallel sum */
    enum { LANES = 512, GROUP = 64 };
    double P[LANES][13];
    memset(P, 0, sizeof(P));
    for (int i = 0; i < F->m; ++i) {
      double t[13];
      functor_terms(F, T, i, t);
      double* A = P[F->si[i] % LANES];
      for (int c = 0; c < 13; ++c) A[c] += t[c];
    }
    for (int g0 = 0; g0 < LANES; g0 += GROUP) {
      for (int off = GROUP / 2; off >= 1; off /= 2)
        for (int l = 0; l < off; ++l)
          for (int c = 0; c < 13; ++c) P[g0 + l][c] += P[g0 + l + off][c];
      for (int c = 0; c < 13; ++c) tot[c] += P[g0][c];
    }
  }
  F->n_evals++;
  const double m = (double)F->m;
  if (f_out) *f_out = tot[0] / m;
  if (g) {
    double Racc[9];
    for (int a = 0; a < 3; ++a) g[a] = tot[1 + a] * (2.0 / m);
    for (int a = 0; a < 9; ++a) Racc[a] = tot[4 + a] * (2.0 / m);
    r_derivative(x, Racc, g);
  }
}

void lo_gicp_cost_grad(const float* src_xyz_guess, const float* tgt_xyz, const int32_t* src_idx,
                       const int32_t* tgt_idx, int m, const double* mahal9, const double x[6],
                       double* f, double g[6]) {
  lo_functor F = {src_xyz_guess, tgt_xyz, src_idx, tgt_idx, m, mahal9, 0, 0, 0};
  functor_fdf(&F, x, f, g);
}

/* ------------------------------------------------------------------ BFGS (pcl::BFGS == GSL vector_bfgs2) */

enum { BF_RUNNING = 0, BF_SUCCESS = 1, BF_NOPROGRESS = 2, BF_NEG = -1 };

typedef struct {
  lo_functor* F;
  double x0[6], g0[6], p[6], dx0[6], dg0[6];
  double f, g[6]; /* current value / gradient */
  double g0norm, pnorm, delta_f, fp0, step;
  /* line-function cache (GSL linear_wrapper.c / PCL BFGS members) */
  double x_alpha[6], g_alpha[6], f_alpha, df_alpha;
  double f_cache_key, df_cache_key, x_cache_key, g_cache_key;
  double rho, sigma, tau1, tau2, tau3;
  int order;
} lo_bfgs;

static double dot6(const double* a, const double* b) {
  double s = 0;
  for (int i = 0; i < 6; ++i) s += a[i] * b[i];
  return s;
}
static double nrm6(const double* a) { return sqrt(dot6(a, a)); }

static void bf_move_to(lo_bfgs* s, double alpha) {
  if (alpha == s->x_cache_key) return;
  for (int i = 0; i < 6; ++i) s->x_alpha[i] = s->x0[i] + alpha * s->p[i];
  s->x_cache_key = alpha;
}
static double bf_slope(lo_bfgs* s) { return dot6(s->g_alpha, s->p); }

static double bf_f(lo_bfgs* s, double alpha) {
  if (alpha == s->f_cache_key) return s->f_alpha;
  bf_move_to(s, alpha);
  functor_fdf(s->F, s->x_alpha, &s->f_alpha, NULL);
  s->f_cache_key = alpha;
  return s->f_alpha;
}
static double bf_df(lo_bfgs* s, double alpha) {
  if (alpha == s->df_cache_key) return s->df_alpha;
  bf_move_to(s, alpha);
  if (alpha != s->g_cache_key) {
    functor_fdf(s->F, s->x_alpha, NULL, s->g_alpha);
    s->g_cache_key = alpha;
  }
  s->df_alpha = bf_slope(s);
  s->df_cache_key = alpha;
  return s->df_alpha;
}
static void bf_fdf(lo_bfgs* s, double alpha, double* f, double* df) {
  if (alpha == s->f_cache_key && alpha == s->df_cache_key) {
    *f = s->f_alpha;
    *df = s->df_alpha;
    return;
  }
  if (alpha == s->f_cache_key || alpha == s->df_cache_key) {
    *f = bf_f(s, alpha);
    *df = bf_df(s, alpha);
    return;
  }
  bf_move_to(s, alpha);
  functor_fdf(s->F, s->x_alpha, &s->f_alpha, s->g_alpha);
  s->f_cache_key = alpha;
  s->g_cache_key = alpha;
  s->df_alpha = bf_slope(s);
  s->df_cache_key = alpha;
  *f = s->f_alpha;
  *df = s->df_alpha;
}
static void bf_update_position(lo_bfgs* s, double alpha, double x[6], double* f, double g[6]) {
  double f_alpha, df_alpha;
  bf_fdf(s, alpha, &f_alpha, &df_alpha);
  *f = f_alpha;
  memcpy(x, s->x_alpha, sizeof(double) * 6);
  memcpy(g, s->g_alpha, sizeof(double) * 6);
}
static void bf_change_direction(lo_bfgs* s) {
  /* convert the cache values from the end of the current minimisation to those needed
   * for the start of the next minimisation, alpha = 0 */
  memcpy(s->x_alpha, s->x0, sizeof(double) * 6);
  s->x_cache_key = 0.0;
  s->f_cache_key = 0.0; /* f_alpha keeps the value at the new x0 */
  memcpy(s->g_alpha, s->g0, sizeof(double) * 6);
  s->g_cache_key = 0.0;
  s->df_alpha = bf_slope(s);
  s->df_cache_key = 0.0;
}

static double cubic(double c0, double c1, double c2, double c3, double z) {
  return c0 + z * (c1 + z * (c2 + z * c3));
}
static void check_extremum(double c0, double c1, double c2, double c3, double z, double* zmin,
                           double* fmin) {
  const double y = cubic(c0, c1, c2, c3, z);
  if (y < *fmin) {
    *zmin = z;
    *fmin = y;
  }
}
/* gsl_poly_solve_quadratic */
static int solve_quadratic(double a, double b, double c, double* x0, double* x1) {
  if (a == 0) {
    if (b == 0) return 0;
    *x0 = -c / b;
    return 1;
  }
  const double disc = b * b - 4 * a * c;
  if (disc > 0) {
    if (b == 0) {
      const double r = sqrt(-c / a);
      *x0 = -r;
      *x1 = r;
    } else {
      const double sgnb = (b > 0 ? 1 : -1);
      const double temp = -0.5 * (b + sgnb * sqrt(disc));
      const double r1 = temp / a, r2 = c / temp;
      if (r1 < r2) {
        *x0 = r1;
        *x1 = r2;
      } else {
        *x0 = r2;
        *x1 = r1;
      }
    }
    return 2;
  } else if (disc == 0) {
    *x0 = -0.5 * b / a;
    *x1 = -0.5 * b / a;
    return 2;
  }
  return 0;
}
static double interp_quad(double f0, double fp0, double f1, double zl, double zh) {
  const double fl = f0 + zl * (fp0 + zl * (f1 - f0 - fp0));
  const double fh = f0 + zh * (fp0 + zh * (f1 - f0 - fp0));
  const double c = 2 * (f1 - f0 - fp0);
  double zmin = zl, fmin = fl;
  if (fh < fmin) {
    zmin = zh;
    fmin = fh;
  }
  if (c > 0) {
    const double z = -fp0 / c;
    if (z > zl && z < zh) {
      const double f = f0 + z * (fp0 + z * (f1 - f0 - fp0));
      if (f < fmin) {
        zmin = z;
        fmin = f;
      }
    }
  }
  return zmin;
}
static double interp_cubic(double f0, double fp0, double f1, double fp1, double zl, double zh) {
  const double eta = 3 * (f1 - f0) - 2 * fp0 - fp1;
  const double xi = fp0 + fp1 - 2 * (f1 - f0);
  const double c0 = f0, c1 = fp0, c2 = eta, c3 = xi;
  double zmin = zl, fmin = cubic(c0, c1, c2, c3, zl), z0 = 0, z1 = 0;
  check_extremum(c0, c1, c2, c3, zh, &zmin, &fmin);
  const int n = solve_quadratic(3 * c3, 2 * c2, c1, &z0, &z1);
  if (n == 2) {
    if (z0 > zl && z0 < zh) check_extremum(c0, c1, c2, c3, z0, &zmin, &fmin);
    if (z1 > zl && z1 < zh) check_extremum(c0, c1, c2, c3, z1, &zmin, &fmin);
  } else if (n == 1) {
    if (z0 > zl && z0 < zh) check_extremum(c0, c1, c2, c3, z0, &zmin, &fmin);
  }
  return zmin;
}
static double interpolate(double a, double fa, double fpa, double b, double fb, double fpb,
                          double xmin, double xmax, int order) {
  double zmin = (xmin - a) / (b - a), zmax = (xmax - a) / (b - a), z;
  if (zmin > zmax) {
    const double t = zmin;
    zmin = zmax;
    zmax = t;
  }
  if (order > 2 && isfinite(fpb))
    z = interp_cubic(fa, fpa * (b - a), fb, fpb * (b - a), zmin, zmax);
  else
    z = interp_quad(fa, fpa * (b - a), fb, zmin, zmax);
  return a + z * (b - a);
}

/* GSL linear_minimize.c: minimize() == pcl::BFGS::lineSearch */
static int line_search(lo_bfgs* s, double alpha1, double* alpha_new) {
  const double rho = s->rho, sigma = s->sigma, tau1 = s->tau1, tau2 = s->tau2, tau3 = s->tau3;
  const int order = s->order;
  double f0, fp0, falpha, falpha_prev, fpalpha = 0, fpalpha_prev, delta, alpha_next;
  double alpha = alpha1, alpha_prev = 0.0;
  double a, b, fa, fb, fpa, fpb;
  const int bracket_iters = 100, section_iters = 100;
  int i = 0;
  bf_fdf(s, 0.0, &f0, &fp0);
  falpha_prev = f0;
  fpalpha_prev = fp0;
  a = 0.0;
  b = alpha;
  fa = f0;
  fb = 0.0;
  fpa = fp0;
  fpb = 0.0;
  while (i++ < bracket_iters) {
    falpha = bf_f(s, alpha);
    if (falpha > f0 + alpha * rho * fp0 || falpha >= falpha_prev) {
      a = alpha_prev;
      fa = falpha_prev;
      fpa = fpalpha_prev;
      b = alpha;
      fb = falpha;
      fpb = NAN;
      break;
    }
    fpalpha = bf_df(s, alpha);
    if (fabs(fpalpha) <= -sigma * fp0) {
      *alpha_new = alpha;
      return BF_SUCCESS;
    }
    if (fpalpha >= 0) {
      a = alpha;
      fa = falpha;
      fpa = fpalpha;
      b = alpha_prev;
      fb = falpha_prev;
      fpb = fpalpha_prev;
      break;
    }
    delta = alpha - alpha_prev;
    {
      const double lower = alpha + delta, upper = alpha + tau1 * delta;
      alpha_next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha,
                               lower, upper, order);
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
      alpha = interpolate(a, fa, fpa, b, fb, fpb, lower, upper, order);
    }
    falpha = bf_f(s, alpha);
    if ((a - alpha) * fpa <= DBL_EPSILON) return BF_NOPROGRESS; /* roundoff prevents progress */
    if (falpha > f0 + rho * alpha * fp0 || falpha >= fa) {
      b = alpha;
      fb = falpha;
      fpb = NAN;
    } else {
      fpalpha = bf_df(s, alpha);
      if (fabs(fpalpha) <= -sigma * fp0) {
        *alpha_new = alpha;
        return BF_SUCCESS;
      }
      if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) {
        b = a;
        fb = fa;
        fpb = fpa;
        a = alpha;
        fa = falpha;
        fpa = fpalpha;
      } else {
        a = alpha;
        fa = falpha;
        fpa = fpalpha;
      }
    }
  }
  return BF_SUCCESS;
}

static void bfgs_init(lo_bfgs* s, lo_functor* F, const double x[6]) {
  memset(s, 0, sizeof(*s));
  s->F = F;
  s->step = 1.0; /* parameters.step_size default */
  s->rho = 0.01;  /* gicp.hpp:253-258 */
  s->sigma = 0.01;
  s->tau1 = 9;
  s->tau2 = 0.05;
  s->tau3 = 0.5;
  s->order = 3;
  s->delta_f = 0;
  functor_fdf(F, x, &s->f, s->g);
  memcpy(s->x0, x, sizeof(double) * 6);
  memcpy(s->g0, s->g, sizeof(double) * 6);
  s->g0norm = nrm6(s->g0);
  for (int i = 0; i < 6; ++i) s->p[i] = s->g[i] * -1 / s->g0norm;
  s->pnorm = nrm6(s->p);
  s->fp0 = -s->g0norm;
  memcpy(s->x_alpha, s->x0, sizeof(double) * 6);
  memcpy(s->g_alpha, s->g0, sizeof(double) * 6);
  s->f_alpha = s->f;
  s->df_alpha = bf_slope(s);
  s->f_cache_key = s->df_cache_key = s->x_cache_key = s->g_cache_key = 0.0;
}

static int bfgs_step(lo_bfgs* s, double x[6]) {
  double alpha = 0.0, alpha1;
  const double f0 = s->f;
  if (s->pnorm == 0.0 || s->g0norm == 0.0 || s->fp0 == 0 || !isfinite(s->fp0)) return BF_NOPROGRESS;
  if (s->delta_f < 0) {
    const double del = fmax(-s->delta_f, 10 * DBL_EPSILON * fabs(f0));
    alpha1 = fmin(1.0, 2.0 * del / (-s->fp0));
  } else {
    alpha1 = fabs(s->step);
  }
  const int status = line_search(s, alpha1, &alpha);
  if (status != BF_SUCCESS) return status;
  bf_update_position(s, alpha, x, &s->f, s->g);
  s->delta_f = s->f - f0;
  {
    for (int i = 0; i < 6; ++i) {
      s->dx0[i] = x[i] - s->x0[i];
      s->dg0[i] = s->g[i] - s->g0[i];
    }
    const double dxg = dot6(s->dx0, s->g), dgg = dot6(s->dg0, s->g);
    const double dxdg = dot6(s->dx0, s->dg0), dgnorm = nrm6(s->dg0);
    double A, B;
    if (dxdg != 0) {
      B = dxg / dxdg;
      A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg;
    } else {
      B = 0;
      A = 0;
    }
    for (int i = 0; i < 6; ++i) s->p[i] = s->g[i] - A * s->dx0[i] - B * s->dg0[i];
  }
  memcpy(s->g0, s->g, sizeof(double) * 6);
  memcpy(s->x0, x, sizeof(double) * 6);
  s->g0norm = nrm6(s->g0);
  s->pnorm = nrm6(s->p);
  const double pg = dot6(s->p, s->g);
  const double dir = (pg >= 0.0) ? -1.0 : +1.0;
  for (int i = 0; i < 6; ++i) s->p[i] *= dir / s->pnorm;
  s->pnorm = nrm6(s->p);
  s->fp0 = dot6(s->p, s->g0);
  bf_change_direction(s);
  return BF_RUNNING; /* BFGSSpace::Success == 0 in PCL == "keep going" for the caller's loop */
}

/* estimateRigidTransformationBFGS (gicp.hpp:217-287). returns 0 ok, -1 exception */
static int estimate_bfgs(lo_functor* F, int max_inner, double gradient_tol, float T[16]) {
  if (F->m < 4) return -1; /* NotEnoughPointsException */
  double x[6];
  x[0] = T[3];
  x[1] = T[7];
  x[2] = T[11];
  x[3] = atan2(T[9], T[10]);
  x[4] = asin(-T[8]);
  x[5] = atan2(T[4], T[0]);
  lo_bfgs s;
  bfgs_init(&s, F, x);
  int inner = 0, result;
  /* gradient_tol: 1e-2 in the reference (gicp.hpp:250) */
  do {
    inner++;
    result = bfgs_step(&s, x);
    if (result) break;
    result = (nrm6(s.g) < gradient_tol) ? BF_SUCCESS : BF_RUNNING; /* testGradient */
  } while (result == BF_RUNNING && inner < max_inner);
  if (result == BF_NOPROGRESS || result == BF_SUCCESS || inner == max_inner) {
    lo_apply_state(x, T); /* setIdentity(); applyState() */
    return 0;
  }
  return -1; /* SolverDidntConvergeException */
}

/* ------------------------------------------------------------------ Gauss-Newton back end
 * Not in the reference: the solver the HIP path runs, restated on the CPU so that the
 * GPU can be checked step by step and the BFGS-vs-GN gap can be measured.  It minimises
 * exactly the reference's objective (gicp.hpp:290-317) for the same fixed correspondences
 * and Mahalanobis matrices.  Because res = Theta*(p,1) - q is linear in the 12 entries of
 * Theta = [R|t], the objective is a quadratic form in delta = vec(Theta(x) - Theta0); the
 * O(m) pass accumulates that form once (Q = sum (u u^T) (x) M, b = sum u (x) M r0,
 * c = sum r0^T M r0, u = (p,1)) and Gauss-Newton then iterates on the 6 Euler parameters
 * at O(1) per step. */
typedef struct {
  double Q[144]; /* index 3*k + a, k = column of Theta (0..3), a = row (0..2) */
  double b[12];
  double c;
  int m;
  double Theta0[12]; /* [k*3 + a] */
} lo_quadform;

static void theta_of_x(const double x[6], double Th[12], double dTh[3][9]) {
  const double phi = x[3], th = x[4], psi = x[5];
  const double cf = cos(phi), sf = sin(phi), ct = cos(th), st = sin(th), cp = cos(psi),
               sp = sin(psi);
  /* R = Rz(psi) Ry(theta) Rx(phi), row-major R[a][k] */
  const double R[9] = {cp * ct, cp * st * sf - sp * cf, cp * st * cf + sp * sf,
                       sp * ct, sp * st * sf + cp * cf, sp * st * cf - cp * sf,
                       -st,     ct * sf,                ct * cf};
  for (int k = 0; k < 3; ++k)
    for (int a = 0; a < 3; ++a) Th[k * 3 + a] = R[a * 3 + k];
  Th[9] = x[0];
  Th[10] = x[1];
  Th[11] = x[2];
  if (dTh) {
    const double dphi[9] = {0, cp * st * cf + sp * sf,  -cp * st * sf + sp * cf,
                            0, sp * st * cf - cp * sf,  -sp * st * sf - cp * cf,
                            0, ct * cf,                 -ct * sf};
    const double dth[9] = {-cp * st, cp * ct * sf, cp * ct * cf,
                           -sp * st, sp * ct * sf, sp * ct * cf,
                           -ct,      -st * sf,     -st * cf};
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

static double quad_eval(const lo_quadform* q, const double Th[12], double* grad12 /*nullable*/) {
  double d[12], Qd[12];
  for (int i = 0; i < 12; ++i) d[i] = Th[i] - q->Theta0[i];
  for (int i = 0; i < 12; ++i) {
    double s = 0;
    for (int j = 0; j < 12; ++j) s += q->Q[i * 12 + j] * d[j];
    Qd[i] = s;
  }
  double f = q->c;
  for (int i = 0; i < 12; ++i) f += d[i] * (2.0 * q->b[i] + Qd[i]);
  if (grad12)
    for (int i = 0; i < 12; ++i) grad12[i] = 2.0 * (q->b[i] + Qd[i]);
  return f;
}

static int chol6_solve(const double H[36], const double g[6], double dx[6]) {
  double L[36] = {0};
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = H[i * 6 + j];
      for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
      if (i == j) {
        if (!(s > 0)) return -1;
        L[i * 6 + i] = sqrt(s);
      } else {
        L[i * 6 + j] = s / L[j * 6 + j];
      }
    }
  double y[6];
  for (int i = 0; i < 6; ++i) {
    double s = g[i];
    for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * y[k];
    y[i] = s / L[i * 6 + i];
  }
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * dx[k];
    dx[i] = s / L[i * 6 + i];
  }
  return 0;
}

static void build_quadform(const lo_functor* F, const float T0[16], lo_quadform* q) {
  memset(q, 0, sizeof(*q));
  q->m = F->m;
  for (int k = 0; k < 3; ++k)
    for (int a = 0; a < 3; ++a) q->Theta0[k * 3 + a] = (double)T0[a * 4 + k];
  for (int a = 0; a < 3; ++a) q->Theta0[9 + a] = (double)T0[a * 4 + 3];
  for (int i = 0; i < F->m; ++i) {
    const float* ps = F->src + 3 * F->si[i];
    const float* pt = F->tgt + 3 * F->ti[i];
    /* residual at Theta0 in double from the float matrix widened to double: the reference
     * rounds T*p to float first (gicp.hpp:307-310); that per-point rounding noise (~1e-6 m)
     * changes whenever T moves by one ulp and keeps the outer loop from ever reaching
     * delta < 1 at tf_epsilon = 1e-10, so the Gauss-Newton back end leaves it out. */
    double r0[3];
    for (int a = 0; a < 3; ++a)
      r0[a] = (q->Theta0[0 * 3 + a] * (double)ps[0] + q->Theta0[1 * 3 + a] * (double)ps[1] +
               q->Theta0[2 * 3 + a] * (double)ps[2] + q->Theta0[9 + a]) -
              (double)pt[a];
    const double* Mi = F->mahal + 9 * (size_t)F->si[i];
    double M[9];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) M[a * 3 + b] = 0.5 * (Mi[a * 3 + b] + Mi[b * 3 + a]);
    const double Mr[3] = {M[0] * r0[0] + M[1] * r0[1] + M[2] * r0[2],
                          M[3] * r0[0] + M[4] * r0[1] + M[5] * r0[2],
                          M[6] * r0[0] + M[7] * r0[1] + M[8] * r0[2]};
    const double u[4] = {ps[0], ps[1], ps[2], 1.0};
    q->c += r0[0] * Mr[0] + r0[1] * Mr[1] + r0[2] * Mr[2];
    for (int k = 0; k < 4; ++k) {
      for (int a = 0; a < 3; ++a) q->b[k * 3 + a] += u[k] * Mr[a];
      for (int l = 0; l < 4; ++l) {
        const double ukl = u[k] * u[l];
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) q->Q[(k * 3 + a) * 12 + (l * 3 + b)] += ukl * M[a * 3 + b];
      }
    }
  }
}

/* Gauss-Newton on the quadratic form; exposed shape shared with the device solver:
 * start at the Euler parameters of T (gicp.hpp:235-241), full steps with step halving when
 * the model cost does not decrease, stop when |dx|_inf < 1e-9, after max_inner steps or
 * when no decrease is possible. */
static int estimate_gn(lo_functor* F, int max_inner, float T[16]) {
  if (F->m < 4) return -1;
  lo_quadform q;
  build_quadform(F, T, &q);
  F->n_evals++;
  double x[6];
  x[0] = T[3];
  x[1] = T[7];
  x[2] = T[11];
  x[3] = atan2(T[9], T[10]);
  x[4] = asin(-T[8]);
  x[5] = atan2(T[4], T[0]);
  double Th[12], dTh[3][9], g12[12];
  theta_of_x(x, Th, dTh);
  double f = quad_eval(&q, Th, g12);
  for (int it = 0; it < max_inner; ++it) {
    /* J (12x6): columns 0..2 translation, 3..5 angles */
    double J[12][6];
    memset(J, 0, sizeof(J));
    for (int a = 0; a < 3; ++a) J[9 + a][a] = 1.0;
    for (int c = 0; c < 3; ++c)
      for (int e = 0; e < 9; ++e) J[e][3 + c] = dTh[c][e];
    double g[6], H[36], QJ[12][6];
    for (int c = 0; c < 6; ++c) {
      double s = 0;
      for (int e = 0; e < 12; ++e) s += J[e][c] * g12[e];
      g[c] = -s; /* right-hand side */
    }
    for (int e = 0; e < 12; ++e)
      for (int c = 0; c < 6; ++c) {
        double s = 0;
        for (int e2 = 0; e2 < 12; ++e2) s += q.Q[e * 12 + e2] * J[e2][c];
        QJ[e][c] = s;
      }
    for (int c = 0; c < 6; ++c)
      for (int c2 = 0; c2 < 6; ++c2) {
        double s = 0;
        for (int e = 0; e < 12; ++e) s += J[e][c] * QJ[e][c2];
        H[c * 6 + c2] = 2.0 * s;
      }
    double dx[6];
    if (chol6_solve(H, g, dx) != 0) break;
    double step = 1.0, fn = f, xn[6];
    int ok = 0;
    for (int h = 0; h < 30; ++h) {
      for (int c = 0; c < 6; ++c) xn[c] = x[c] + step * dx[c];
      theta_of_x(xn, Th, NULL);
      fn = quad_eval(&q, Th, NULL);
      if (fn < f) {
        ok = 1;
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
    theta_of_x(x, Th, dTh);
    f = quad_eval(&q, Th, g12);
    if (mx < 1e-9) break;
  }
  lo_apply_state(x, T);
  return 0;
}

/* ------------------------------------------------------------------ computeTransformation */

#define LO_CYCLE_HISTORY 16
/* FNV-1a over the 16 float words of a transformation */
static uint64_t hash_T(const float T[16]) {
  uint64_t h = 1469598103934665603ull;
  for (int i = 0; i < 16; ++i) {
    uint32_t w;
    memcpy(&w, &T[i], 4);
    for (int b = 0; b < 4; ++b) {
      h ^= (w >> (8 * b)) & 0xFFu;
      h *= 1099511628211ull;
    }
  }
  return h;
}

void lo_gicp_default_params(lo_gicp_params* p) {
  p->corr_dist = 5.0;              /* gicp.h:135 */
  p->tf_epsilon = 5e-4;            /* gicp.h:134 */
  p->rotation_epsilon = 2e-3;      /* gicp.h:128 */
  p->max_iterations = 200;         /* gicp.h:133 */
  p->max_inner_iterations = 20;    /* gicp.h:130 */
  p->k_correspondences = 20;       /* gicp.h:121 */
  p->gicp_epsilon = 0.001;         /* gicp.h:127 */
  p->cov_from_knn = 0;
  p->solver = LO_SOLVER_BFGS;
  p->use_kdtree = 1;
  p->threads = 1;
  p->gradient_tol = 1e-2;          /* gicp.hpp:250 */
  p->gn_cycle_exit = 1;
  p->residual_double = 0;
  p->sum_order = 0;
}

int lo_gicp_align(const lo_gicp_params* p, const float* src_xyz, const float* src_normals,
                  int n_src, const float* tgt_xyz, const float* tgt_normals, int n_tgt,
                  const float guess[16], lo_gicp_result* out, int32_t* trace_idx, float* trace_T,
                  int trace_cap) {
  memset(out, 0, sizeof(*out));
  const int N = n_src;
  double* mahal = (double*)malloc(sizeof(double) * 9 * (size_t)(N > 0 ? N : 1));
  for (int i = 0; i < N; ++i) { /* mahalanobis_.resize(N, Identity) gicp.hpp:418 */
    double* M = mahal + 9 * (size_t)i;
    for (int a = 0; a < 9; ++a) M[a] = (a % 4 == 0) ? 1.0 : 0.0;
  }
  double* covT = (double*)malloc(sizeof(double) * 9 * (size_t)(n_tgt > 0 ? n_tgt : 1));
  double* covS = (double*)malloc(sizeof(double) * 9 * (size_t)(N > 0 ? N : 1));
  int cov_ok = 1;
  /* computeCovariances (gicp.hpp:64-156): both branches bail out when the cloud is
   * smaller than k_correspondences_, leaving the covariance vector empty */
  if (p->k_correspondences > n_tgt || p->k_correspondences > N) cov_ok = 0;
  if (cov_ok) {
    if (!p->cov_from_knn) {
      lo_cov_from_normals(tgt_normals, n_tgt, covT);
      lo_cov_from_normals(src_normals, N, covS);
    } else {
      lo_cov_knn(tgt_xyz, n_tgt, p->k_correspondences, p->gicp_epsilon, covT, p->threads);
      lo_cov_knn(src_xyz, N, p->k_correspondences, p->gicp_epsilon, covS, p->threads);
    }
  }
  float T[16], Tprev[16], Tfinal[16];
  for (int i = 0; i < 16; ++i) T[i] = Tprev[i] = (i % 5 == 0) ? 1.f : 0.f;
  float* output = (float*)malloc(sizeof(float) * 3 * (size_t)(N > 0 ? N : 1));
  lo_transform_points_f32(guess, src_xyz, N, output); /* gicp.hpp:440 */
  float* query = (float*)malloc(sizeof(float) * 3 * (size_t)(N > 0 ? N : 1));
  int32_t* nn_idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
  float* nn_d = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
  int32_t* si = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
  int32_t* ti = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
  lo_kdtree* tree = p->use_kdtree ? lo_kdtree_build(tgt_xyz, n_tgt) : NULL;
  const double dist_threshold = p->corr_dist * p->corr_dist;
  int converged = 0, nr_iterations = 0;
  double delta = 0.;
  uint64_t hist[LO_CYCLE_HISTORY]; /* hashes of earlier lookup transformations (GN cycle exit) */
  memset(hist, 0, sizeof(hist));
  if (!cov_ok) {
    /* the reference would index an empty covariance vector here (undefined behaviour);
     * the oracle reports a failed, unconverged alignment instead */
    out->failure = 1;
  }
  while (!converged && cov_ok) {
    /* transform_R = transformation_ * guess in double (gicp.hpp:450-460) */
    double R[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double s = 0;
        for (int k = 0; k < 4; ++k) s += (double)T[i * 4 + k] * (double)guess[k * 4 + j];
        R[i * 3 + j] = s;
      }
    for (int i = 0; i < N; ++i) xform4(T, output + 3 * i, query + 3 * i); /* gicp.hpp:468-469 */
    if (tree)
      lo_kdtree_nn(tree, query, N, nn_idx, nn_d, p->threads);
    else
      lo_nn_brute(query, N, tgt_xyz, n_tgt, nn_idx, nn_d);
    int m = 0;
    for (int i = 0; i < N; ++i) {
      int32_t tj = -1;
      if (nn_idx[i] >= 0 && (double)nn_d[i] < dist_threshold) { /* gicp.hpp:483 */
        const double* C1 = covS + 9 * (size_t)i;
        const double* C2 = covT + 9 * (size_t)nn_idx[i];
        double RC[9], tmp[9];
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += R[a * 3 + k] * C1[k * 3 + b];
            RC[a * 3 + b] = s;
          }
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += RC[a * 3 + k] * R[b * 3 + k];
            tmp[a * 3 + b] = s + C2[a * 3 + b];
          }
        lo_mat3_inverse(tmp, mahal + 9 * (size_t)i);
        si[m] = i;
        ti[m] = nn_idx[i];
        ++m;
        tj = nn_idx[i];
      }
      if (trace_idx && nr_iterations < trace_cap) trace_idx[(size_t)nr_iterations * N + i] = tj;
    }
    if (trace_T && nr_iterations < trace_cap)
      memcpy(trace_T + 16 * (size_t)nr_iterations, T, sizeof(float) * 16);
    out->last_num_correspondences = m;
    memcpy(Tprev, T, sizeof(T)); /* previous_transformation_ = transformation_ */
    lo_functor F = {output, tgt_xyz, si, ti, m, mahal, 0, p->residual_double, p->sum_order};
    const int rc = (p->solver == LO_SOLVER_GN)
                       ? estimate_gn(&F, p->max_inner_iterations, T)
                       : estimate_bfgs(&F, p->max_inner_iterations,
                                       p->gradient_tol > 0 ? p->gradient_tol : 1e-2, T);
    if (rc != 0) break; /* catch (PCLException) -> break, converged_ stays false */
    delta = 0.;
    for (int k = 0; k < 4; ++k)
      for (int l = 0; l < 4; ++l) {
        const double ratio = (k < 3 && l < 3) ? 1. / p->rotation_epsilon : 1. / p->tf_epsilon;
        const double c_delta = ratio * fabs((double)(Tprev[k * 4 + l] - T[k * 4 + l]));
        if (c_delta > delta) delta = c_delta;
      }
    nr_iterations++;
    /* Gauss-Newton back end only (not in the reference): with tf_epsilon = 1e-10 the test
     * delta < 1 asks for a bit-identical float translation, and the exact minimiser can settle
     * into a short limit cycle in which one correspondence flips and T alternates between
     * float-adjacent matrices (periods 2 and 6 have been observed).  A transformation that was
     * already used for the lookups within the last LO_CYCLE_HISTORY iterations proves that no
     * further progress is possible: stop as converged. */
    int cycle = 0;
    if (p->solver == LO_SOLVER_GN && p->gn_cycle_exit) {
      const uint64_t h = hash_T(T);
      const int n_hist = nr_iterations - 1 < LO_CYCLE_HISTORY ? nr_iterations - 1 : LO_CYCLE_HISTORY;
      for (int k = 0; k < n_hist; ++k)
        if (hist[k] == h) cycle = 1;
      /* ring of the transformations used for the lookups of the last iterations */
      hist[(nr_iterations - 1) % LO_CYCLE_HISTORY] = hash_T(Tprev);
    }
    if (nr_iterations >= p->max_iterations || delta < 1 || cycle) {
      converged = 1;
      out->cycle_exit = cycle && !(nr_iterations >= p->max_iterations || delta < 1);
      memcpy(Tprev, T, sizeof(T));
    }
  }
  lo_mat4f_mul(Tprev, guess, Tfinal); /* gicp.hpp:583 */
  memcpy(out->final_transformation, Tfinal, sizeof(Tfinal));
  out->converged = converged;
  out->iterations = nr_iterations;
  out->last_delta = delta;
  if (tree) lo_kdtree_free(tree);
  free(mahal);
  free(covT);
  free(covS);
  free(output);
  free(query);
  free(nn_idx);
  free(nn_d);
  free(si);
  free(ti);
  return 0;
}

/* Measurement helper, not a reference function: the reference's objective (gicp.hpp:290-402) and
 * its gradient at the stationary point an alignment claims.  T plays the part of the guess, the
 * lookups and Mahalanobis matrices are built as in one outer iteration starting from
 * transformation_ = identity (gicp.hpp:445-498), and the functor is evaluated at x = 0.  A
 * transformation at the fixed point of the outer loop has a vanishing gradient here. */
int lo_gicp_objective_at(const lo_gicp_params* p, const float* src_xyz, const float* src_normals,
                         int n_src, const float* tgt_xyz, const float* tgt_normals, int n_tgt,
                         const float T[16], double* f, double g[6], int* m_out) {
  const int N = n_src;
  if (N <= 0 || n_tgt <= 0) return -1;
  double* mahal = (double*)malloc(sizeof(double) * 9 * (size_t)N);
  double* covT = (double*)malloc(sizeof(double) * 9 * (size_t)n_tgt);
  double* covS = (double*)malloc(sizeof(double) * 9 * (size_t)N);
  float* output = (float*)malloc(sizeof(float) * 3 * (size_t)N);
  int32_t* nn_idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
  float* nn_d = (float*)malloc(sizeof(float) * (size_t)N);
  int32_t* si = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
  int32_t* ti = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
  lo_cov_from_normals(tgt_normals, n_tgt, covT);
  lo_cov_from_normals(src_normals, N, covS);
  lo_transform_points_f32(T, src_xyz, N, output);
  lo_kdtree* tree = lo_kdtree_build(tgt_xyz, n_tgt);
  lo_kdtree_nn(tree, output, N, nn_idx, nn_d, p->threads);
  lo_kdtree_free(tree);
  double R[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = (double)T[i * 4 + j];
  const double dist_threshold = p->corr_dist * p->corr_dist;
  int m = 0;
  for (int i = 0; i < N; ++i) {
    if (nn_idx[i] < 0 || !((double)nn_d[i] < dist_threshold)) continue;
    const double* C1 = covS + 9 * (size_t)i;
    const double* C2 = covT + 9 * (size_t)nn_idx[i];
    double RC[9], tmp[9];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += R[a * 3 + k] * C1[k * 3 + b];
        RC[a * 3 + b] = s;
      }
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += RC[a * 3 + k] * R[b * 3 + k];
        tmp[a * 3 + b] = s + C2[a * 3 + b];
      }
    lo_mat3_inverse(tmp, mahal + 9 * (size_t)i);
    si[m] = i;
    ti[m] = nn_idx[i];
    ++m;
  }
  int rc = -1;
  if (m >= 4) {
    lo_functor F = {output, tgt_xyz, si, ti, m, mahal, 0, 1, 0};
    const double x0[6] = {0, 0, 0, 0, 0, 0};
    functor_fdf(&F, x0, f, g);
    rc = 0;
  }
  if (m_out) *m_out = m;
  free(mahal);
  free(covT);
  free(covS);
  free(output);
  free(nn_idx);
  free(nn_d);
  free(si);
  free(ti);
  return rc;
}

/* pcl::Registration::getFitnessScore(max_range = DBL_MAX) + the correspondence pass of
 * IcpLoopComputation.cc:553-566 (same queries, same tree) */
double lo_fitness_score(const float* src_xyz, int n_src, const float* tgt_xyz, int n_tgt,
                        const float final_T[16], int32_t* idx_out, int use_kdtree, int threads) {
  float* q = (float*)malloc(sizeof(float) * 3 * (size_t)(n_src > 0 ? n_src : 1));
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_src > 0 ? n_src : 1));
  float* d = (float*)malloc(sizeof(float) * (size_t)(n_src > 0 ? n_src : 1));
  lo_transform_points_f32(final_T, src_xyz, n_src, q);
  if (use_kdtree) {
    lo_kdtree* tree = lo_kdtree_build(tgt_xyz, n_tgt);
    lo_kdtree_nn(tree, q, n_src, idx, d, threads);
    lo_kdtree_free(tree);
  } else {
    lo_nn_brute(q, n_src, tgt_xyz, n_tgt, idx, d);
  }
  double s = 0.0;
  int nr = 0;
  for (int i = 0; i < n_src; ++i) {
    if (idx[i] < 0) continue;
    if ((double)d[i] <= DBL_MAX) {
      s += d[i];
      nr++;
    }
  }
  if (idx_out) memcpy(idx_out, idx, sizeof(int32_t) * (size_t)n_src);
  free(q);
  free(idx);
  free(d);
  return nr > 0 ? s / nr : DBL_MAX;
}
