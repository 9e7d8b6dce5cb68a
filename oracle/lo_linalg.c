/*
 * lo_linalg.c — small dense algebra for the CPU oracle (TEST INFRASTRUCTURE ONLY, see
 * lamp_oracle.h).  Restates the Eigen routines the reference calls on tiny fixed-size
 * matrices; Eigen itself is not under /root/reference, so each routine follows Eigen's
 * published algorithm and is pinned only through the reference's end-to-end fixtures.
 */
#include "lamp_oracle.h"

#include <math.h>
#include <string.h>

/* Eigen::Matrix3d::inverse(): cofactors + 1/det, det = cofactor column 0 . column 0
 * (call site: lamp_utils/include/lamp_utils/gicp.hpp:493). */
void lo_mat3_inverse(const double a[9], double out[9]) {
  const double a00 = a[0], a01 = a[1], a02 = a[2];
  const double a10 = a[3], a11 = a[4], a12 = a[5];
  const double a20 = a[6], a21 = a[7], a22 = a[8];
  const double c00 = a11 * a22 - a12 * a21;
  const double c10 = a12 * a20 - a10 * a22; /* cofactor(1,0) of the transpose layout */
  const double c20 = a10 * a21 - a11 * a20;
  const double det = a00 * c00 + a01 * c10 + a02 * c20;
  const double invdet = 1.0 / det;
  out[0] = c00 * invdet;
  out[1] = (a02 * a21 - a01 * a22) * invdet;
  out[2] = (a01 * a12 - a02 * a11) * invdet;
  out[3] = c10 * invdet;
  out[4] = (a00 * a22 - a02 * a20) * invdet;
  out[5] = (a02 * a10 - a00 * a12) * invdet;
  out[6] = c20 * invdet;
  out[7] = (a01 * a20 - a00 * a21) * invdet;
  out[8] = (a00 * a11 - a01 * a10) * invdet;
}

/* LU with partial pivoting (what Eigen's general inverse()/determinant() use for 6x6;
 * call sites: loop_closure/src/IcpLoopComputation.cc:725,732,883). */
static int lu6(double m[36], int perm[6], int* sign) {
  *sign = 1;
  for (int i = 0; i < 6; ++i) perm[i] = i;
  for (int k = 0; k < 6; ++k) {
    int piv = k;
    double best = fabs(m[k * 6 + k]);
    for (int i = k + 1; i < 6; ++i) {
      double v = fabs(m[i * 6 + k]);
      if (v > best) {
        best = v;
        piv = i;
      }
    }
    if (best == 0.0) return -1;
    if (piv != k) {
      for (int j = 0; j < 6; ++j) {
        double tmp = m[k * 6 + j];
        m[k * 6 + j] = m[piv * 6 + j];
        m[piv * 6 + j] = tmp;
      }
      int t = perm[k];
      perm[k] = perm[piv];
      perm[piv] = t;
      *sign = -*sign;
    }
    for (int i = k + 1; i < 6; ++i) {
      m[i * 6 + k] /= m[k * 6 + k];
      const double l = m[i * 6 + k];
      for (int j = k + 1; j < 6; ++j) m[i * 6 + j] -= l * m[k * 6 + j];
    }
  }
  return 0;
}

double lo_mat6_det(const double a[36]) {
  double m[36];
  int perm[6], sign;
  memcpy(m, a, sizeof(m));
  if (lu6(m, perm, &sign) != 0) return 0.0;
  double d = sign;
  for (int i = 0; i < 6; ++i) d *= m[i * 6 + i];
  return d;
}

int lo_mat6_inverse(const double a[36], double out[36]) {
  double m[36];
  int perm[6], sign;
  memcpy(m, a, sizeof(m));
  if (lu6(m, perm, &sign) != 0) {
    for (int i = 0; i < 36; ++i) out[i] = NAN;
    return -1;
  }
  for (int c = 0; c < 6; ++c) {
    double y[6];
    for (int i = 0; i < 6; ++i) {
      double s = (perm[i] == c) ? 1.0 : 0.0;
      for (int j = 0; j < i; ++j) s -= m[i * 6 + j] * y[j];
      y[i] = s;
    }
    for (int i = 5; i >= 0; --i) {
      double s = y[i];
      for (int j = i + 1; j < 6; ++j) s -= m[i * 6 + j] * out[j * 6 + c];
      out[i * 6 + c] = s / m[i * 6 + i];
    }
  }
  return 0;
}

/* Cyclic Jacobi for a symmetric n x n matrix (n <= 6).  Stands in for
 * Eigen::SelfAdjointEigenSolver (PointCloudUtils.cc:189), Eigen::JacobiSVD of a symmetric
 * PSD 3x3 (gicp.hpp:140, singular vectors == eigenvectors) and Eigen::EigenSolver of the
 * (numerically symmetric) covariance at IcpLoopComputation.cc:886-905.
 * evals ascending; evecs row-major, column k is the k-th eigenvector. */
void lo_sym_eig(const double* a_in, int n, double* evals, double* evecs) {
  double a[36], v[36];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      a[i * n + j] = 0.5 * (a_in[i * n + j] + a_in[j * n + i]);
      v[i * n + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        if (i == j)
          diag += a[i * n + j] * a[i * n + j];
        else
          off += a[i * n + j] * a[i * n + j];
      }
    if (off <= 1e-32 * (diag + off) || off == 0.0) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[p * n + q];
        if (apq == 0.0) continue;
        const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = a[k * n + p], akq = a[k * n + q];
          a[k * n + p] = c * akp - s * akq;
          a[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = a[p * n + k], aqk = a[q * n + k];
          a[p * n + k] = c * apk - s * aqk;
          a[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = v[k * n + p], vkq = v[k * n + q];
          v[k * n + p] = c * vkp - s * vkq;
          v[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  int order[6];
  for (int i = 0; i < n; ++i) order[i] = i;
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j)
      if (a[order[j] * n + order[j]] < a[order[i] * n + order[i]]) {
        int t = order[i];
        order[i] = order[j];
        order[j] = t;
      }
  for (int k = 0; k < n; ++k) {
    evals[k] = a[order[k] * n + order[k]];
    if (evecs)
      for (int i = 0; i < n; ++i) evecs[i * n + k] = v[i * n + order[k]];
  }
}

/* Eigen::LDLT<Matrix6d, Lower> in-place algorithm with diagonal pivoting
 * (call site: IcpLoopComputation.cc:737-739).  The reference later recomposes
 * L * D * L^T WITHOUT the transpositions (:757); callers restate that. */
void lo_ldlt(const double a_in[36], double L[36], double D[6], int transp[6]) {
  double A[36];
  memcpy(A, a_in, sizeof(A));
  const int n = 6;
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double best = fabs(A[k * n + k]);
    for (int i = k + 1; i < n; ++i) {
      const double v = fabs(A[i * n + i]);
      if (v > best) {
        best = v;
        piv = i;
      }
    }
    transp[k] = piv;
    if (piv != k) {
      /* symmetric swap on the lower triangle, as Eigen does */
      for (int j = 0; j < k; ++j) {
        double t = A[k * n + j];
        A[k * n + j] = A[piv * n + j];
        A[piv * n + j] = t;
      }
      for (int i = piv + 1; i < n; ++i) {
        double t = A[i * n + k];
        A[i * n + k] = A[i * n + piv];
        A[i * n + piv] = t;
      }
      {
        double t = A[k * n + k];
        A[k * n + k] = A[piv * n + piv];
        A[piv * n + piv] = t;
      }
      for (int i = k + 1; i < piv; ++i) {
        double t = A[i * n + k];
        A[i * n + k] = A[piv * n + i];
        A[piv * n + i] = t;
      }
    }
    double temp[6];
    if (k > 0) {
      for (int j = 0; j < k; ++j) temp[j] = A[j * n + j] * A[k * n + j];
      double s = 0.0;
      for (int j = 0; j < k; ++j) s += A[k * n + j] * temp[j];
      A[k * n + k] -= s;
      for (int i = k + 1; i < n; ++i) {
        double s2 = 0.0;
        for (int j = 0; j < k; ++j) s2 += A[i * n + j] * temp[j];
        A[i * n + k] -= s2;
      }
    }
    const double akk = A[k * n + k];
    if (fabs(akk) > 0.0)
      for (int i = k + 1; i < n; ++i) A[i * n + k] /= akk;
  }
  for (int i = 0; i < n; ++i) {
    D[i] = A[i * n + i];
    for (int j = 0; j < n; ++j) L[i * n + j] = (j < i) ? A[i * n + j] : (i == j ? 1.0 : 0.0);
  }
}

/* ------------------------------------------------------------------ rigid-body helpers */

/* Eigen::Quaterniond(w,x,y,z).toRotationMatrix() without normalisation — what
 * gtsam::Rot3(w,x,y,z) stores (call sites: IcpLoopComputation.cc:409-413,
 * lamp_utils/include/lamp_utils/CommonFunctions.h:171-179). */
void lo_pose_from_quat(const double q[4], const double t[3], lo_pose* out) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  out->R[0] = 1 - (tyy + tzz);
  out->R[1] = txy - twz;
  out->R[2] = txz + twy;
  out->R[3] = txy + twz;
  out->R[4] = 1 - (txx + tzz);
  out->R[5] = tyz - twx;
  out->R[6] = txz - twy;
  out->R[7] = tyz + twx;
  out->R[8] = 1 - (txx + tyy);
  out->t[0] = t[0];
  out->t[1] = t[1];
  out->t[2] = t[2];
}

/* gtsam::Pose3::between: a^-1 * b  (IcpLoopComputation.cc:496,533,597,923) */
void lo_pose_between(const lo_pose* a, const lo_pose* b, lo_pose* out) {
  lo_pose r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += a->R[k * 3 + i] * b->R[k * 3 + j];
      r.R[i * 3 + j] = s;
    }
  for (int i = 0; i < 3; ++i) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += a->R[k * 3 + i] * (b->t[k] - a->t[k]);
    r.t[i] = s;
  }
  *out = r;
}

/* geometry_utils::PoseInverse (IcpLoopComputation.cc:617): (R^T, -R^T t) */
void lo_pose_inverse(const lo_pose* a, lo_pose* out) {
  lo_pose r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.R[i * 3 + j] = a->R[j * 3 + i];
  for (int i = 0; i < 3; ++i) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += a->R[k * 3 + i] * a->t[k];
    r.t[i] = -s;
  }
  *out = r;
}

/* gtsam::Pose3::Logmap (IcpLoopComputation.cc:599): xi = (omega, u), GTSAM's published
 * formulas for SO(3) log and the SE(3) translation part. */
void lo_pose_logmap(const lo_pose* p, double xi[6]) {
  const double* R = p->R;
  const double tr = R[0] + R[4] + R[8];
  double w[3];
  if (tr + 1.0 < 1e-10) {
    /* rotation by pi: pick the numerically safest axis */
    if (fabs(R[8] + 1.0) > 1e-5) {
      const double s = M_PI / sqrt(2.0 + 2.0 * R[8]);
      w[0] = s * R[2];
      w[1] = s * R[5];
      w[2] = s * (1.0 + R[8]);
    } else if (fabs(R[4] + 1.0) > 1e-5) {
      const double s = M_PI / sqrt(2.0 + 2.0 * R[4]);
      w[0] = s * R[1];
      w[1] = s * (1.0 + R[4]);
      w[2] = s * R[7];
    } else {
      const double s = M_PI / sqrt(2.0 + 2.0 * R[0]);
      w[0] = s * (1.0 + R[0]);
      w[1] = s * R[3];
      w[2] = s * R[6];
    }
  } else {
    double magnitude;
    const double tr_3 = tr - 3.0;
    if (tr_3 < -1e-7) {
      const double theta = acos((tr - 1.0) / 2.0);
      magnitude = theta / (2.0 * sin(theta));
    } else {
      magnitude = 0.5 - tr_3 * tr_3 / 12.0;
    }
    w[0] = magnitude * (R[7] - R[5]);
    w[1] = magnitude * (R[2] - R[6]);
    w[2] = magnitude * (R[3] - R[1]);
  }
  const double t = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  xi[0] = w[0];
  xi[1] = w[1];
  xi[2] = w[2];
  if (t < 1e-10) {
    xi[3] = p->t[0];
    xi[4] = p->t[1];
    xi[5] = p->t[2];
    return;
  }
  const double k[3] = {w[0] / t, w[1] / t, w[2] / t};
  const double* T = p->t;
  /* W = skew(k) */
  const double WT[3] = {k[1] * T[2] - k[2] * T[1], k[2] * T[0] - k[0] * T[2],
                        k[0] * T[1] - k[1] * T[0]};
  const double WWT[3] = {k[1] * WT[2] - k[2] * WT[1], k[2] * WT[0] - k[0] * WT[2],
                         k[0] * WT[1] - k[1] * WT[0]};
  const double Tan = tan(0.5 * t);
  for (int i = 0; i < 3; ++i) xi[3 + i] = T[i] - (0.5 * t) * WT[i] + (1 - t / (2. * Tan)) * WWT[i];
}
