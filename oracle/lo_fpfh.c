/*
 * lo_fpfh.c — CPU restatement of the FPFH descriptors behind lamp_utils::ComputeFeatures
 * (TEST INFRASTRUCTURE ONLY, see lamp_oracle.h).
 *
 * Call site: lamp_utils/src/PointCloudUtils.cc:98-113 (pcl::FPFHEstimationOMP with
 * input = keypoints, search surface = full cloud, radius search).  PCL is not under
 * /root/reference; the algorithm below is PCL's published FPFHEstimation
 * (computePairFeatures, computePointSPFHSignature, weightPointSPFHSignature) with 11+11+11
 * bins.  No reference fixture pins descriptor values (its test only checks the count,
 * lamp_utils/test/test_point_cloud_utils.cc:153-170): parity unpinned.
 */
#include "lamp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NB 11

static float dist2f(const float* a, const float* b) {
  float d = a[0] - b[0];
  float r = d * d;
  d = a[1] - b[1];
  r = r + d * d;
  d = a[2] - b[2];
  r = r + d * d;
  return r;
}

/* pcl::computePairFeatures */
static int pair_features(const float* p1, const float* n1, const float* p2, const float* n2,
                         float* f1, float* f2, float* f3, float* f4) {
  float d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  *f4 = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (*f4 == 0.0f) {
    *f1 = *f2 = *f3 = *f4 = 0.0f;
    return 0;
  }
  float a[3] = {n1[0], n1[1], n1[2]}, b[3] = {n2[0], n2[1], n2[2]};
  const float angle1 = (a[0] * d[0] + a[1] * d[1] + a[2] * d[2]) / *f4;
  const float angle2 = (b[0] * d[0] + b[1] * d[1] + b[2] * d[2]) / *f4;
  if (acosf(fabsf(angle1)) > acosf(fabsf(angle2))) {
    for (int k = 0; k < 3; ++k) {
      const float t = a[k];
      a[k] = b[k];
      b[k] = t;
      d[k] *= -1.f;
    }
    *f3 = -angle2;
  } else {
    *f3 = angle1;
  }
  float v[3] = {d[1] * a[2] - d[2] * a[1], d[2] * a[0] - d[0] * a[2], d[0] * a[1] - d[1] * a[0]};
  const float v_norm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (v_norm == 0.0f) {
    *f1 = *f2 = *f3 = *f4 = 0.0f;
    return 0;
  }
  const float s = 1.0f / v_norm;
  v[0] *= s;
  v[1] *= s;
  v[2] *= s;
  const float w[3] = {a[1] * v[2] - a[2] * v[1], a[2] * v[0] - a[0] * v[2],
                      a[0] * v[1] - a[1] * v[0]};
  *f2 = v[0] * b[0] + v[1] * b[1] + v[2] * b[2];
  *f1 = atan2f(w[0] * b[0] + w[1] * b[1] + w[2] * b[2], a[0] * b[0] + a[1] * b[1] + a[2] * b[2]);
  return 1;
}

static int bin_of(float x, int nb) {
  int h = (int)floorf(x);
  if (h < 0) h = 0;
  if (h >= nb) h = nb - 1;
  return h;
}

/* FPFH at `keypoints` over `surface` (exhaustive radius search, neighbours in ascending
 * surface index).  out is [n_key][33] = [f1 | f2 | f3]. */
void lo_fpfh(const float* surface_xyz, const float* surface_normals, int n,
             const float* keypoints_xyz, int n_key, double radius, float* out) {
  const float r2 = (float)(radius * radius);
  unsigned char* need = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
  for (int k = 0; k < n_key; ++k)
    for (int j = 0; j < n; ++j)
      if (dist2f(keypoints_xyz + 3 * k, surface_xyz + 3 * j) < r2) need[j] = 1;
  float* spfh = (float*)calloc((size_t)(n > 0 ? n : 1) * 33, sizeof(float));
  int* nbr = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) {
    if (!need[i]) continue;
    int cnt = 0;
    for (int j = 0; j < n; ++j)
      if (dist2f(surface_xyz + 3 * i, surface_xyz + 3 * j) < r2) nbr[cnt++] = j;
    if (cnt == 0) continue;
    /* computePointSPFHSignature */
    const float hist_incr = 100.0f / (float)(cnt - 1);
    float* h = spfh + (size_t)i * 33;
    for (int q = 0; q < cnt; ++q) {
      const int j = nbr[q];
      if (j == i) continue;
      float f1, f2, f3, f4;
      if (!pair_features(surface_xyz + 3 * i, surface_normals + 3 * i, surface_xyz + 3 * j,
                         surface_normals + 3 * j, &f1, &f2, &f3, &f4))
        continue;
      h[bin_of((float)(NB * ((f1 + M_PI) * (1.0 / (2.0 * M_PI)))), NB)] += hist_incr;
      h[NB + bin_of((float)(NB * ((f2 + 1.0) * 0.5)), NB)] += hist_incr;
      h[2 * NB + bin_of((float)(NB * ((f3 + 1.0) * 0.5)), NB)] += hist_incr;
    }
  }
  for (int k = 0; k < n_key; ++k) {
    float* o = out + (size_t)k * 33;
    for (int b = 0; b < 33; ++b) o[b] = 0.f;
    double sum[3] = {0, 0, 0};
    /* weightPointSPFHSignature */
    for (int j = 0; j < n; ++j) {
      const float d2 = dist2f(keypoints_xyz + 3 * k, surface_xyz + 3 * j);
      if (!(d2 < r2)) continue; /* FLANN RadiusResultSet admits dist < radius^2 only */
      if (d2 == 0) continue;
      const float weight = 1.0f / d2;
      for (int b = 0; b < 33; ++b) {
        const float val = spfh[(size_t)j * 33 + b] * weight;
        sum[b / NB] += val;
        o[b] += val;
      }
    }
    for (int g = 0; g < 3; ++g) {
      if (sum[g] != 0) sum[g] = 100.0 / sum[g];
      for (int b = 0; b < NB; ++b) o[g * NB + b] *= (float)sum[g];
    }
  }
  free(need);
  free(spfh);
  free(nbr);
}
