/*
 * lo_features.c — CPU restatement of the feature front end of the FEATURES initialisation:
 * Harris3D keypoints (lamp_utils::ComputeKeypoints with normals,
 * lamp_utils/src/PointCloudUtils.cc:78-96) and SAC-IA
 * (IcpLoopComputation::GetSacInitialAlignment, loop_closure/src/IcpLoopComputation.cc:650-705).
 * TEST INFRASTRUCTURE ONLY, see lamp_oracle.h.
 *
 * Both are PCL classes (pcl::HarrisKeypoint3D, pcl::SampleConsensusInitialAlignment) that
 * are not under /root/reference; what follows restates PCL's published algorithms.  Reference
 * pins: three keypoints on the corner fixture, the first at the origin
 * (lamp_utils/test/test_point_cloud_utils.cc:135-151); SAC-IA recovers the corner shift exactly
 * with fitness 0 (loop_closure/test/test_loop_computation.cc:78-99).  PCL draws its samples
 * with rand(); here a fixed LCG does, so only distributional agreement is possible.
 */
#include "lamp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static float d2f(const float* a, const float* b) {
  float d = a[0] - b[0];
  float r = d * d;
  d = a[1] - b[1];
  r = r + d * d;
  d = a[2] - b[2];
  r = r + d * d;
  return r;
}

/* HarrisKeypoint3D::responseHarris: covariance of the normals in the radius neighbourhood,
 * response = 0.04 + det - 0.04 trace^2 (float), neighbours visited in ascending index. */
void lo_harris_response(const float* xyz, const float* normals, int n, double radius,
                        float* response) {
  const float r2 = (float)(radius * radius);
  for (int i = 0; i < n; ++i) {
    response[i] = 0.f;
    if (!isfinite(xyz[3 * i]) || !isfinite(xyz[3 * i + 1]) || !isfinite(xyz[3 * i + 2])) continue;
    float c[8] = {0, 0, 0, 0, 0, 0, 0, 0}, zz = 0.f;
    unsigned count = 0;
    for (int j = 0; j < n; ++j) {
      if (!(d2f(xyz + 3 * i, xyz + 3 * j) < r2)) continue;
      const float* m = normals + 3 * j;
      if (!isfinite(m[0])) continue;
      c[0] += m[0] * m[0];
      c[1] += m[1] * m[0];
      c[2] += m[2] * m[0];
      c[4] += m[0] * m[1];
      c[5] += m[1] * m[1];
      c[6] += m[2] * m[1];
      zz += m[2] * m[2];
      ++count;
    }
    if (count > 0) {
      for (int k = 0; k < 8; ++k) c[k] = c[k] / (float)count;
      c[7] = zz / (float)count;
    }
    const float trace = c[0] + c[5] + c[7];
    if (trace != 0) {
      const float det = c[0] * c[5] * c[7] + 2.0f * c[1] * c[2] * c[6] - c[2] * c[2] * c[5] -
                        c[1] * c[1] * c[7] - c[6] * c[6] * c[0];
      response[i] = 0.04f + det - 0.04f * trace * trace;
    }
  }
}

/* detectKeypoints with non-maxima suppression, no refinement; keypoints in ascending index
 * (PCL appends them from an OpenMP loop, so its order is not defined).  Returns the count. */
int lo_harris_keypoints(const float* xyz, const float* normals, int n, double radius,
                        double threshold, int nms, int32_t* out_idx, float* out_response) {
  float* resp = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  lo_harris_response(xyz, normals, n, radius, resp);
  const float r2 = (float)(radius * radius);
  const float thr = (float)threshold;
  int k = 0;
  for (int i = 0; i < n; ++i) {
    if (!nms) {
      out_idx[k] = i;
      if (out_response) out_response[k] = resp[i];
      ++k;
      continue;
    }
    if (!isfinite(xyz[3 * i]) || !isfinite(resp[i]) || resp[i] < thr) continue;
    int is_max = 1;
    for (int j = 0; j < n && is_max; ++j)
      if (d2f(xyz + 3 * i, xyz + 3 * j) < r2 && resp[i] < resp[j]) is_max = 0;
    if (is_max) {
      out_idx[k] = i;
      if (out_response) out_response[k] = resp[i];
      ++k;
    }
  }
  free(resp);
  return k;
}

/* ------------------------------------------------------------------ SAC-IA */
static uint32_t lcg_next(uint32_t* s) {
  *s = *s * 1664525u + 1013904223u;
  return (*s >> 8) & 0x7FFFFFu;
}

/* TransformationEstimationSVD for (few) correspondences: R = V U^T of H = sum s t^T */
static void rigid_from_pairs(const float* src, const int* si, const float* tgt, const int* ti,
                             int m, float T[16]) {
  double cs[3] = {0, 0, 0}, ct[3] = {0, 0, 0};
  for (int i = 0; i < m; ++i)
    for (int a = 0; a < 3; ++a) {
      cs[a] += src[3 * si[i] + a];
      ct[a] += tgt[3 * ti[i] + a];
    }
  for (int a = 0; a < 3; ++a) {
    cs[a] /= m;
    ct[a] /= m;
  }
  double H[9] = {0};
  for (int i = 0; i < m; ++i)
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b)
        H[a * 3 + b] += (src[3 * si[i] + a] - cs[a]) * (tgt[3 * ti[i] + b] - ct[b]);
  /* V = eigenvectors of H^T H (descending), u_i = H v_i normalised, third vectors by cross */
  double HtH[9], ev[3], V[9];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += H[k * 3 + a] * H[k * 3 + b];
      HtH[a * 3 + b] = s;
    }
  lo_sym_eig(HtH, 3, ev, V); /* ascending: column 2 is the largest */
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const double s1 = sqrt(ev[2] > 0 ? ev[2] : 0), s2 = sqrt(ev[1] > 0 ? ev[1] : 0);
  double scale = 0;
  for (int a = 0; a < 3; ++a) scale += cs[a] * cs[a] + ct[a] * ct[a];
  scale = 1e-12 * (1.0 + scale);
  if (s1 > scale) {
    double v1[3] = {V[2], V[5], V[8]}, v2[3] = {V[1], V[4], V[7]}, v3[3];
    double u1[3], u2[3], u3[3];
    for (int a = 0; a < 3; ++a) {
      u1[a] = (H[a * 3] * v1[0] + H[a * 3 + 1] * v1[1] + H[a * 3 + 2] * v1[2]) / s1;
      u2[a] = H[a * 3] * v2[0] + H[a * 3 + 1] * v2[1] + H[a * 3 + 2] * v2[2];
    }
    const double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    for (int a = 0; a < 3; ++a) u1[a] /= n1;
    if (s2 > 1e-9 * s1) {
      /* rank >= 2 (the generic case for three samples): third vectors by cross product,
       * which also makes det(R) = +1 (PCL flips V's last column to the same effect) */
      const double dp = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
      for (int a = 0; a < 3; ++a) u2[a] -= dp * u1[a];
      const double n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
      for (int a = 0; a < 3; ++a) u2[a] /= n2;
      u3[0] = u1[1] * u2[2] - u1[2] * u2[1];
      u3[1] = u1[2] * u2[0] - u1[0] * u2[2];
      u3[2] = u1[0] * u2[1] - u1[1] * u2[0];
      v3[0] = v1[1] * v2[2] - v1[2] * v2[1];
      v3[1] = v1[2] * v2[0] - v1[0] * v2[2];
      v3[2] = v1[0] * v2[1] - v1[1] * v2[0];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) R[a * 3 + b] = v1[a] * u1[b] + v2[a] * u2[b] + v3[a] * u3[b];
    } else {
      /* rank 1 (collinear samples): the smallest rotation that turns u1 onto v1 */
      double ax[3] = {u1[1] * v1[2] - u1[2] * v1[1], u1[2] * v1[0] - u1[0] * v1[2],
                      u1[0] * v1[1] - u1[1] * v1[0]};
      const double sn = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
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
  } /* else: coincident samples, H = 0: JacobiSVD returns U = V = I, so R = I */
  for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.f : 0.f;
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) T[a * 4 + b] = (float)R[a * 3 + b];
    T[a * 4 + 3] = (float)(ct[a] - (R[a * 3] * cs[0] + R[a * 3 + 1] * cs[1] + R[a * 3 + 2] * cs[2]));
  }
}

static float error_metric(const float* cloud, int n, const float* tgt, int nt, float threshold) {
  /* PCL adds the per-point errors into a float in point order; the device sums them as a
   * tree in double and rounds once, and the oracle follows the device so that both take the
   * same "strictly lower" decisions */
  double error = 0;
  for (int i = 0; i < n; ++i) {
    float best = INFINITY;
    for (int j = 0; j < nt; ++j) {
      const float d = d2f(cloud + 3 * i, tgt + 3 * j);
      if (d < best) best = d;
    }
    error += (double)((best <= threshold) ? best / threshold : 1.0f); /* TruncatedError */
  }
  return (float)error;
}

/* pcl::SampleConsensusInitialAlignment::computeTransformation with nr_samples 3,
 * min_sample_distance 0, corr_dist_threshold 100 (its constructor default),
 * k_correspondences as given (IcpLoopComputation.cc:697 sets 5).  Features are [n][33].
 * Returns the fitness score (mean squared NN distance of the aligned source keypoints),
 * DBL_MAX when fewer than 3 source keypoints or no target keypoint exist. */
double lo_sac_ia(const float* src_kp, const float* src_feat, int n_src, const float* tgt_kp,
                 const float* tgt_feat, int n_tgt, int max_iterations, int k_corr,
                 const float guess[16], uint32_t seed, float T_out[16]) {
  memcpy(T_out, guess, sizeof(float) * 16);
  if (n_src < 3 || n_tgt < 1) return DBL_MAX;
  const float thr = 100.0f;
  float* moved = (float*)malloc(sizeof(float) * 3 * (size_t)n_src);
  float lowest = 0.f;
  int i_iter = 0, is_identity = 1;
  for (int i = 0; i < 16; ++i)
    if (guess[i] != ((i % 5 == 0) ? 1.f : 0.f)) is_identity = 0;
  if (!is_identity) {
    lo_transform_points_f32(guess, src_kp, n_src, moved);
    lowest = error_metric(moved, n_src, tgt_kp, n_tgt, thr);
    i_iter = 1;
  }
  const int k = k_corr < n_tgt ? k_corr : n_tgt;
  int32_t* nn = (int32_t*)malloc(sizeof(int32_t) * (size_t)k);
  float* nd = (float*)malloc(sizeof(float) * (size_t)k);
  uint32_t rng = seed;
  for (; i_iter < max_iterations; ++i_iter) {
    int samples[3], n_s = 0, corr[3];
    while (n_s < 3) { /* selectSamples: unique indices (min_sample_distance = 0) */
      const int s = (int)(lcg_next(&rng) % (uint32_t)n_src);
      int ok = 1;
      for (int q = 0; q < n_s; ++q)
        if (samples[q] == s) ok = 0;
      if (ok) samples[n_s++] = s;
    }
    for (int q = 0; q < 3; ++q) { /* findSimilarFeatures */
      for (int p = 0; p < k; ++p) {
        nd[p] = INFINITY;
        nn[p] = -1;
      }
      const float* f = src_feat + 33 * (size_t)samples[q];
      for (int j = 0; j < n_tgt; ++j) {
        float d = 0;
        for (int b = 0; b < 33; ++b) {
          const float e = f[b] - tgt_feat[33 * (size_t)j + b];
          d += e * e;
        }
        if (d < nd[k - 1]) {
          int p = k - 1;
          while (p > 0 && d < nd[p - 1]) {
            nd[p] = nd[p - 1];
            nn[p] = nn[p - 1];
            --p;
          }
          nd[p] = d;
          nn[p] = j;
        }
      }
      corr[q] = nn[lcg_next(&rng) % (uint32_t)k];
    }
    float T[16];
    rigid_from_pairs(src_kp, samples, tgt_kp, corr, 3, T);
    lo_transform_points_f32(T, src_kp, n_src, moved);
    const float err = error_metric(moved, n_src, tgt_kp, n_tgt, thr);
    if (i_iter == 0 || err < lowest) {
      lowest = err;
      memcpy(T_out, T, sizeof(T));
    }
  }
  lo_transform_points_f32(T_out, src_kp, n_src, moved);
  double fit = 0;
  for (int i = 0; i < n_src; ++i) {
    float best = INFINITY;
    for (int j = 0; j < n_tgt; ++j) {
      const float d = d2f(moved + 3 * i, tgt_kp + 3 * j);
      if (d < best) best = d;
    }
    fit += best;
  }
  free(moved);
  free(nn);
  free(nd);
  return fit / n_src;
}
