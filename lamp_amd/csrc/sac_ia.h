// sac_ia.h — host side of the FEATURES initialisation: SAC-IA over Harris keypoints and their
// FPFH descriptors (IcpLoopComputation::GetSacInitialAlignment,
// loop_closure/src/IcpLoopComputation.cc:650-705 -> pcl::SampleConsensusInitialAlignment).
//
// Split of the work: the descriptor k-NN table and the scoring of every hypothesis against the
// keypoint clouds run on the GPU (feature_init.hip); drawing the samples and estimating one
// 3-point rigid transform per hypothesis is O(1) scalar work per hypothesis and stays on the
// host.  Algorithm (PCL's published one):
// every iteration draws 3 distinct source keypoints, gives each a random one of its k nearest
// target descriptors, estimates the rigid transform of the three pairs (SVD of the 3x3
// correlation), and scores it by sum over source keypoints of min(d^2 / 100, 1), d^2 = squared
// distance to the nearest target keypoint; the lowest score wins; a non-identity guess is
// scored first as hypothesis 0.  PCL samples with rand(); a fixed LCG is used here so that a
// run is reproducible.
#pragma once

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace lgicp {
namespace sac {

inline uint32_t lcg_next(uint32_t& s) {
  s = s * 1664525u + 1013904223u;
  return (s >> 8) & 0x7FFFFFu;
}

inline float d2f(const float* a, const float* b) {
  float d = a[0] - b[0];
  float r = d * d;
  d = a[1] - b[1];
  r = r + d * d;
  d = a[2] - b[2];
  r = r + d * d;
  return r;
}

inline void transform(const float T[16], const float* in, int n, float* out) {
  for (int i = 0; i < n; ++i) {
    const float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
    out[3 * i + 0] = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
    out[3 * i + 1] = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
    out[3 * i + 2] = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
  }
}

// symmetric 3x3 eigen-decomposition (cyclic Jacobi), ascending eigenvalues, vectors in columns
inline void eig3(const double A_in[9], double ev[3], double V[9]) {
  double A[9];
  for (int i = 0; i < 9; ++i) A[i] = A_in[i], V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; ++sweep) {
    const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    const double diag = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off == 0.0 || 2 * off <= 1e-32 * (diag + 2 * off)) break;
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      const double apq = A[p * 3 + q];
      if (apq == 0.0) continue;
      const double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
      const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
      const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
      for (int r = 0; r < 3; ++r) {
        const double arp = A[r * 3 + p], arq = A[r * 3 + q];
        A[r * 3 + p] = c * arp - s * arq, A[r * 3 + q] = s * arp + c * arq;
      }
      for (int r = 0; r < 3; ++r) {
        const double apr = A[p * 3 + r], aqr = A[q * 3 + r];
        A[p * 3 + r] = c * apr - s * aqr, A[q * 3 + r] = s * apr + c * aqr;
      }
      for (int r = 0; r < 3; ++r) {
        const double vrp = V[r * 3 + p], vrq = V[r * 3 + q];
        V[r * 3 + p] = c * vrp - s * vrq, V[r * 3 + q] = s * vrp + c * vrq;
      }
    }
  }
  int o[3] = {0, 1, 2};
  for (int i = 0; i < 3; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (A[o[j] * 4] < A[o[i] * 4]) std::swap(o[i], o[j]);
  double Vs[9];
  for (int k = 0; k < 3; ++k) {
    ev[k] = A[o[k] * 4];
    for (int r = 0; r < 3; ++r) Vs[r * 3 + k] = V[r * 3 + o[k]];
  }
  std::memcpy(V, Vs, sizeof(Vs));
}

// pcl TransformationEstimationSVD on m index pairs (R = V U^T of H = sum s t^T, det fixed)
inline void rigid_from_pairs(const float* src, const int* si, const float* tgt, const int* ti, int m,
                             float T[16]) {
  double cs[3] = {0, 0, 0}, ct[3] = {0, 0, 0};
  for (int i = 0; i < m; ++i)
    for (int a = 0; a < 3; ++a) cs[a] += src[3 * si[i] + a], ct[a] += tgt[3 * ti[i] + a];
  for (int a = 0; a < 3; ++a) cs[a] /= m, ct[a] /= m;
  double H[9] = {0};
  for (int i = 0; i < m; ++i)
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b)
        H[a * 3 + b] += (src[3 * si[i] + a] - cs[a]) * (tgt[3 * ti[i] + b] - ct[b]);
  double HtH[9], ev[3], V[9];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += H[k * 3 + a] * H[k * 3 + b];
      HtH[a * 3 + b] = s;
    }
  eig3(HtH, ev, V);
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const double s1 = std::sqrt(ev[2] > 0 ? ev[2] : 0), s2 = std::sqrt(ev[1] > 0 ? ev[1] : 0);
  double scale = 0;
  for (int a = 0; a < 3; ++a) scale += cs[a] * cs[a] + ct[a] * ct[a];
  scale = 1e-12 * (1.0 + scale);
  if (s1 > scale) {
    double v1[3] = {V[2], V[5], V[8]}, v2[3] = {V[1], V[4], V[7]}, v3[3], u1[3], u2[3], u3[3];
    for (int a = 0; a < 3; ++a) {
      u1[a] = (H[a * 3] * v1[0] + H[a * 3 + 1] * v1[1] + H[a * 3 + 2] * v1[2]) / s1;
      u2[a] = H[a * 3] * v2[0] + H[a * 3 + 1] * v2[1] + H[a * 3 + 2] * v2[2];
    }
    const double n1 = std::sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    for (int a = 0; a < 3; ++a) u1[a] /= n1;
    if (s2 > 1e-9 * s1) {
      const double dp = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
      for (int a = 0; a < 3; ++a) u2[a] -= dp * u1[a];
      const double n2 = std::sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
      for (int a = 0; a < 3; ++a) u2[a] /= n2;
      u3[0] = u1[1] * u2[2] - u1[2] * u2[1], u3[1] = u1[2] * u2[0] - u1[0] * u2[2],
      u3[2] = u1[0] * u2[1] - u1[1] * u2[0];
      v3[0] = v1[1] * v2[2] - v1[2] * v2[1], v3[1] = v1[2] * v2[0] - v1[0] * v2[2],
      v3[2] = v1[0] * v2[1] - v1[1] * v2[0];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) R[a * 3 + b] = v1[a] * u1[b] + v2[a] * u2[b] + v3[a] * u3[b];
    } else {  // collinear samples: the smallest rotation that turns u1 onto v1
      double ax[3] = {u1[1] * v1[2] - u1[2] * v1[1], u1[2] * v1[0] - u1[0] * v1[2],
                      u1[0] * v1[1] - u1[1] * v1[0]};
      const double sn = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
      const double cn = u1[0] * v1[0] + u1[1] * v1[1] + u1[2] * v1[2];
      if (sn > 1e-12) {
        for (int a = 0; a < 3; ++a) ax[a] /= sn;
        const double K[9] = {0, -ax[2], ax[1], ax[2], 0, -ax[0], -ax[1], ax[0], 0};
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) {
            double kk = 0;
            for (int k = 0; k < 3; ++k) kk += K[a * 3 + k] * K[k * 3 + b];
            R[a * 3 + b] = (a == b ? 1.0 : 0.0) + sn * K[a * 3 + b] + (1 - cn) * kk;
          }
      }
    }
  }  // coincident samples (H = 0): the SVD factors are identities, R = I
  for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.f : 0.f;
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) T[a * 4 + b] = (float)R[a * 3 + b];
    T[a * 4 + 3] = (float)(ct[a] - (R[a * 3] * cs[0] + R[a * 3 + 1] * cs[1] + R[a * 3 + 2] * cs[2]));
  }
}

struct Hypotheses {
  std::vector<float> T;  // [n][16]
  int first = 0;         // index of the first RANSAC hypothesis (1 when the guess is scored as #0)
};

// Everything of the RANSAC loop that does not look at the clouds: sample triples, pick one of
// the k nearest target descriptors for each sample (nn is [n_src][k], ascending distance),
// estimate the rigid transform.  Hypothesis 0 is the guess itself when it is not the identity.
inline Hypotheses make_hypotheses(const float* src_kp, int n_src, const float* tgt_kp,
                                  const int* nn, int k, int max_iterations, const float guess[16],
                                  uint32_t seed) {
  Hypotheses h;
  bool identity = true;
  for (int i = 0; i < 16; ++i) identity = identity && guess[i] == ((i % 5 == 0) ? 1.f : 0.f);
  int it = 0;
  if (!identity) {
    h.T.insert(h.T.end(), guess, guess + 16);
    it = 1;
    h.first = 1;
  }
  uint32_t rng = seed;
  for (; it < max_iterations; ++it) {
    int samples[3], n_s = 0, corr[3];
    while (n_s < 3) {
      const int s = (int)(lcg_next(rng) % (uint32_t)n_src);
      bool ok = true;
      for (int q = 0; q < n_s; ++q) ok = ok && samples[q] != s;
      if (ok) samples[n_s++] = s;
    }
    for (int q = 0; q < 3; ++q) corr[q] = nn[(size_t)samples[q] * k + lcg_next(rng) % (uint32_t)k];
    float T[16];
    rigid_from_pairs(src_kp, samples, tgt_kp, corr, 3, T);
    h.T.insert(h.T.end(), T, T + 16);
  }
  return h;
}

}  // namespace sac
}  // namespace lgicp
