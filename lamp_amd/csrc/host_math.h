// host_math.h — O(1)-per-pair host algebra of the loop-closure engine: rigid-body poses,
// 6x6 inverse / LDL^T / symmetric eigen-decomposition.  Everything per-point runs on the GPU;
// what is here is the handful of 6x6 operations per verified pair that the reference also does
// once per pair (loop_closure/src/IcpLoopComputation.cc:597-640, 707-768, 883-905).
#pragma once

#include <cmath>
#include <cstring>

namespace lgicp {
namespace hm {

struct Pose {
  double R[9];
  double t[3];
};

// gtsam::Rot3(w, x, y, z): Eigen quaternion -> matrix without normalisation, so the all-zero
// quaternion of a default-constructed geometry_msgs/Pose maps to the identity
// (IcpLoopComputation.cc:409-413)
inline Pose pose_from_quat(const double q[4], const double t[3]) {
  Pose p;
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  p.R[0] = 1 - (tyy + tzz);
  p.R[1] = txy - twz;
  p.R[2] = txz + twy;
  p.R[3] = txy + twz;
  p.R[4] = 1 - (txx + tzz);
  p.R[5] = tyz - twx;
  p.R[6] = txz - twy;
  p.R[7] = tyz + twx;
  p.R[8] = 1 - (txx + tyy);
  p.t[0] = t[0];
  p.t[1] = t[1];
  p.t[2] = t[2];
  return p;
}

// gtsam Pose3::between: a^-1 * b
inline Pose between(const Pose& a, const Pose& b) {
  Pose r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += a.R[k * 3 + i] * b.R[k * 3 + j];
      r.R[i * 3 + j] = s;
    }
  for (int i = 0; i < 3; ++i) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += a.R[k * 3 + i] * (b.t[k] - a.t[k]);
    r.t[i] = s;
  }
  return r;
}

// geometry_utils::PoseInverse
inline Pose inverse(const Pose& a) {
  Pose r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.R[i * 3 + j] = a.R[j * 3 + i];
  for (int i = 0; i < 3; ++i) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += a.R[k * 3 + i] * a.t[k];
    r.t[i] = -s;
  }
  return r;
}

// gtsam Pose3::Logmap -> (omega, u)
inline void logmap(const Pose& p, double xi[6]) {
  const double* R = p.R;
  const double tr = R[0] + R[4] + R[8];
  double w[3];
  if (tr + 1.0 < 1e-10) {
    if (std::fabs(R[8] + 1.0) > 1e-5) {
      const double s = M_PI / std::sqrt(2.0 + 2.0 * R[8]);
      w[0] = s * R[2], w[1] = s * R[5], w[2] = s * (1.0 + R[8]);
    } else if (std::fabs(R[4] + 1.0) > 1e-5) {
      const double s = M_PI / std::sqrt(2.0 + 2.0 * R[4]);
      w[0] = s * R[1], w[1] = s * (1.0 + R[4]), w[2] = s * R[7];
    } else {
      const double s = M_PI / std::sqrt(2.0 + 2.0 * R[0]);
      w[0] = s * (1.0 + R[0]), w[1] = s * R[3], w[2] = s * R[6];
    }
  } else {
    double mag;
    const double tr3 = tr - 3.0;
    if (tr3 < -1e-7) {
      const double th = std::acos((tr - 1.0) / 2.0);
      mag = th / (2.0 * std::sin(th));
    } else {
      mag = 0.5 - tr3 * tr3 / 12.0;
    }
    w[0] = mag * (R[7] - R[5]), w[1] = mag * (R[2] - R[6]), w[2] = mag * (R[3] - R[1]);
  }
  const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  xi[0] = w[0], xi[1] = w[1], xi[2] = w[2];
  if (th < 1e-10) {
    xi[3] = p.t[0], xi[4] = p.t[1], xi[5] = p.t[2];
    return;
  }
  const double k[3] = {w[0] / th, w[1] / th, w[2] / th};
  const double* T = p.t;
  const double WT[3] = {k[1] * T[2] - k[2] * T[1], k[2] * T[0] - k[0] * T[2],
                        k[0] * T[1] - k[1] * T[0]};
  const double WWT[3] = {k[1] * WT[2] - k[2] * WT[1], k[2] * WT[0] - k[0] * WT[2],
                         k[0] * WT[1] - k[1] * WT[0]};
  const double tn = std::tan(0.5 * th);
  for (int i = 0; i < 3; ++i) xi[3 + i] = T[i] - (0.5 * th) * WT[i] + (1 - th / (2. * tn)) * WWT[i];
}

// rotation matrix -> unit quaternion (w, x, y, z), w >= 0
inline void quat_from_R(const double R[9], double q[4]) {
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    const double s = std::sqrt(tr + 1.0) * 2;
    q[0] = 0.25 * s, q[1] = (R[7] - R[5]) / s, q[2] = (R[2] - R[6]) / s, q[3] = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const double s = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
    q[0] = (R[7] - R[5]) / s, q[1] = 0.25 * s, q[2] = (R[1] + R[3]) / s, q[3] = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    const double s = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
    q[0] = (R[2] - R[6]) / s, q[1] = (R[1] + R[3]) / s, q[2] = 0.25 * s, q[3] = (R[5] + R[7]) / s;
  } else {
    const double s = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
    q[0] = (R[3] - R[1]) / s, q[1] = (R[2] + R[6]) / s, q[2] = (R[5] + R[7]) / s, q[3] = 0.25 * s;
  }
  if (q[0] < 0)
    for (int i = 0; i < 4; ++i) q[i] = -q[i];
}

// ---- 6x6 ------------------------------------------------------------------------------------
struct LU6 {
  double m[36];
  int perm[6];
  int sign;
  bool singular;
};

inline LU6 lu6(const double a[36]) {
  LU6 f;
  std::memcpy(f.m, a, sizeof(f.m));
  f.sign = 1;
  f.singular = false;
  for (int i = 0; i < 6; ++i) f.perm[i] = i;
  for (int k = 0; k < 6; ++k) {
    int piv = k;
    double best = std::fabs(f.m[k * 6 + k]);
    for (int i = k + 1; i < 6; ++i)
      if (std::fabs(f.m[i * 6 + k]) > best) best = std::fabs(f.m[i * 6 + k]), piv = i;
    if (best == 0.0) {
      f.singular = true;
      return f;
    }
    if (piv != k) {
      for (int j = 0; j < 6; ++j) std::swap(f.m[k * 6 + j], f.m[piv * 6 + j]);
      std::swap(f.perm[k], f.perm[piv]);
      f.sign = -f.sign;
    }
    for (int i = k + 1; i < 6; ++i) {
      const double l = (f.m[i * 6 + k] /= f.m[k * 6 + k]);
      for (int j = k + 1; j < 6; ++j) f.m[i * 6 + j] -= l * f.m[k * 6 + j];
    }
  }
  return f;
}

inline double det6(const double a[36]) {
  const LU6 f = lu6(a);
  if (f.singular) return 0.0;
  double d = f.sign;
  for (int i = 0; i < 6; ++i) d *= f.m[i * 6 + i];
  return d;
}

inline bool inverse6(const double a[36], double out[36]) {
  const LU6 f = lu6(a);
  if (f.singular) {
    for (int i = 0; i < 36; ++i) out[i] = NAN;
    return false;
  }
  for (int c = 0; c < 6; ++c) {
    double y[6];
    for (int i = 0; i < 6; ++i) {
      double s = (f.perm[i] == c) ? 1.0 : 0.0;
      for (int j = 0; j < i; ++j) s -= f.m[i * 6 + j] * y[j];
      y[i] = s;
    }
    for (int i = 5; i >= 0; --i) {
      double s = y[i];
      for (int j = i + 1; j < 6; ++j) s -= f.m[i * 6 + j] * out[j * 6 + c];
      out[i * 6 + c] = s / f.m[i * 6 + i];
    }
  }
  return true;
}

// LDL^T with diagonal pivoting, lower-triangular in-place form (the algorithm of Eigen::LDLT,
// which IcpLoopComputation.cc:737-739 calls).  L is unit lower, D the pivoted diagonal.
inline void ldlt6(const double a[36], double L[36], double D[6]) {
  double A[36];
  std::memcpy(A, a, sizeof(A));
  const int n = 6;
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double best = std::fabs(A[k * n + k]);
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(A[i * n + i]) > best) best = std::fabs(A[i * n + i]), piv = i;
    if (piv != k) {
      for (int j = 0; j < k; ++j) std::swap(A[k * n + j], A[piv * n + j]);
      for (int i = piv + 1; i < n; ++i) std::swap(A[i * n + k], A[i * n + piv]);
      std::swap(A[k * n + k], A[piv * n + piv]);
      for (int i = k + 1; i < piv; ++i) std::swap(A[i * n + k], A[piv * n + i]);
    }
    if (k > 0) {
      double temp[6];
      for (int j = 0; j < k; ++j) temp[j] = A[j * n + j] * A[k * n + j];
      double s = 0;
      for (int j = 0; j < k; ++j) s += A[k * n + j] * temp[j];
      A[k * n + k] -= s;
      for (int i = k + 1; i < n; ++i) {
        double s2 = 0;
        for (int j = 0; j < k; ++j) s2 += A[i * n + j] * temp[j];
        A[i * n + k] -= s2;
      }
    }
    const double akk = A[k * n + k];
    if (std::fabs(akk) > 0)
      for (int i = k + 1; i < n; ++i) A[i * n + k] /= akk;
  }
  for (int i = 0; i < n; ++i) {
    D[i] = A[i * n + i];
    for (int j = 0; j < n; ++j) L[i * n + j] = j < i ? A[i * n + j] : (i == j ? 1.0 : 0.0);
  }
}

// cyclic Jacobi on the symmetric part; eigenvalues ascending, eigenvectors in columns
inline void sym_eig6(const double a[36], double ev[6], double V[36]) {
  const int n = 6;
  double A[36];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      A[i * n + j] = 0.5 * (a[i * n + j] + a[j * n + i]);
      V[i * n + j] = i == j ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = 0, diag = 0;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) (i == j ? diag : off) += A[i * n + j] * A[i * n + j];
    if (off == 0.0 || off <= 1e-32 * (diag + off)) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[p * n + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq, A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk, A[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - s * vkq, V[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  int order[6];
  for (int i = 0; i < n; ++i) order[i] = i;
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j)
      if (A[order[j] * n + order[j]] < A[order[i] * n + order[i]]) std::swap(order[i], order[j]);
  double Vs[36];
  for (int k = 0; k < n; ++k) {
    ev[k] = A[order[k] * n + order[k]];
    for (int i = 0; i < n; ++i) Vs[i * n + k] = V[i * n + order[k]];
  }
  std::memcpy(V, Vs, sizeof(Vs));
}

inline void fixed_covariance(double rot_sigma, double trans_sigma, double cov[36]) {
  for (int i = 0; i < 36; ++i) cov[i] = 0.0;
  for (int i = 0; i < 3; ++i) cov[i * 6 + i] = rot_sigma * rot_sigma;
  for (int i = 3; i < 6; ++i) cov[i * 6 + i] = trans_sigma * trans_sigma;
}

// tail of ComputeICPCovariancePointPlane (IcpLoopComputation.cc:724-767) given Ap
inline void point_plane_covariance(const double Ap[36], double rot_sigma, double trans_sigma,
                                   double cov[36]) {
  if (det6(Ap) == 0) {
    fixed_covariance(rot_sigma, trans_sigma, cov);
    return;
  }
  double inv[36];
  inverse6(Ap, inv);
  for (int i = 0; i < 36; ++i) cov[i] = 0.05 * 0.05 * inv[i];
  double L[36], D[6];
  ldlt6(cov, L, D);
  bool recompute = false;
  for (int i = 0; i < 6; ++i) {
    if (D[i] <= 0) D[i] = 1e-12, recompute = true;
    if (D[i] > 1) D[i] = 1, recompute = true;
  }
  if (recompute)  // the reference recomposes without undoing the pivoting (:757)
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        double s = 0;
        for (int k = 0; k < 6; ++k) s += L[i * 6 + k] * D[k] * L[j * 6 + k];
        cov[i * 6 + j] = s;
      }
  for (int i = 0; i < 36; ++i)
    if (std::isnan(cov[i])) {
      fixed_covariance(rot_sigma, trans_sigma, cov);
      return;
    }
}

// tail of ComputeICPCovariancePointPoint (IcpLoopComputation.cc:883-905) given H = sum J^T J:
// cov = H^-1 * fitness, eigenvalues clamped to [1e-3, 1e3].  The reference uses the general
// Eigen::EigenSolver and recomposes V L V^-1; for this (numerically symmetric) matrix that is
// the symmetric decomposition V L V^T.
inline void point_point_covariance(const double H[36], double fitness, double cov[36]) {
  double inv[36];
  inverse6(H, inv);
  bool finite = true;
  for (int i = 0; i < 36; ++i) {
    cov[i] = inv[i] * fitness;
    finite = finite && std::isfinite(cov[i]);
  }
  if (!finite) {  // "Failed to find eigen values": identity * upper bound (:892-896)
    for (int i = 0; i < 36; ++i) cov[i] = (i % 7 == 0) ? 1000.0 : 0.0;
    return;
  }
  double ev[6], V[36];
  sym_eig6(cov, ev, V);
  for (int i = 0; i < 6; ++i) ev[i] = ev[i] < 0.001 ? 0.001 : (ev[i] > 1000 ? 1000 : ev[i]);
  for (int a = 0; a < 6; ++a)
    for (int b = 0; b < 6; ++b) {
      double s = 0;
      for (int k = 0; k < 6; ++k) s += V[a * 6 + k] * ev[k] * V[b * 6 + k];
      cov[a * 6 + b] = s;
    }
}

}  // namespace hm
}  // namespace lgicp
