/* lo_normals.c — CPU restatement of lamp_utils::ComputeNormals / ExtractNormals
 * (lamp_utils/include/lamp_utils/PointCloudUtils.h:38-62, lamp_utils/src/PointCloudUtils.cc:17-35)
 * = pcl::NormalEstimationOMP with a k-nearest-neighbour search (k = 10 by default,
 * factor_handlers/config/normals_computation.yaml).  TEST INFRASTRUCTURE ONLY.
 *
 * PCL is not part of the reference tree; its published algorithm (PCL 1.8, float arithmetic)
 * is restated: the k nearest points of the cloud itself (the query included, ascending
 * distance), their single-pass mean and second moments (computeMeanAndCovarianceMatrix), the
 * eigenvector of the smallest eigenvalue by the closed-form cubic roots + cross products of
 * pcl::eigen33, curvature = |lambda_0 / trace|, and the flip towards the viewpoint (0,0,0).
 * The reference pins plane normals only (test_point_cloud_utils.cc:60-85); other values are
 * parity unpinned. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "lamp_oracle.h"

static void roots2(float b, float c, float roots[3]) {
  roots[0] = 0.f;
  float d = b * b - 4.0f * c;
  if (d < 0.0f) d = 0.0f;
  const float sd = sqrtf(d);
  roots[2] = 0.5f * (b + sd);
  roots[1] = 0.5f * (b - sd);
}

static void swapf(float* a, float* b) {
  const float t = *a;
  *a = *b;
  *b = t;
}

/* pcl::computeRoots for a symmetric 3x3 (row-major m) */
static void compute_roots(const float m[9], float roots[3]) {
  const float c0 = m[0] * m[4] * m[8] + 2.0f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] -
                   m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
  const float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
  const float c2 = m[0] + m[4] + m[8];
  if (fabsf(c0) < 1.1920929e-07f) {
    roots2(c2, c1, roots);
    return;
  }
  const float s_inv3 = 1.0f / 3.0f;
  const float s_sqrt3 = sqrtf(3.0f);
  const float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.0f) a_over_3 = 0.0f;
  const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
  float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.0f) q = 0.0f;
  const float rho = sqrtf(-a_over_3);
  const float theta = atan2f(sqrtf(-q), half_b) * s_inv3;
  const float cos_theta = cosf(theta), sin_theta = sinf(theta);
  roots[0] = c2_over_3 + 2.0f * rho * cos_theta;
  roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  if (roots[0] >= roots[1]) swapf(&roots[0], &roots[1]);
  if (roots[1] >= roots[2]) {
    swapf(&roots[1], &roots[2]);
    if (roots[0] >= roots[1]) swapf(&roots[0], &roots[1]);
  }
  if (roots[0] <= 0.0f) roots2(c2, c1, roots); /* a PSD matrix has no negative eigenvalue */
}

static void cross3(const float* a, const float* b, float* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

/* pcl::eigen33(mat, eigenvalue, eigenvector): smallest eigenvalue and its eigenvector */
static void eigen33_smallest(const float mat[9], float* eigenvalue, float vec[3]) {
  float scale = 0.f;
  for (int i = 0; i < 9; ++i)
    if (fabsf(mat[i]) > scale) scale = fabsf(mat[i]);
  if (scale <= 1.17549435e-38f) scale = 1.0f;
  float s[9];
  for (int i = 0; i < 9; ++i) s[i] = mat[i] / scale;
  float roots[3];
  compute_roots(s, roots);
  *eigenvalue = roots[0] * scale;
  s[0] -= roots[0];
  s[4] -= roots[0];
  s[8] -= roots[0];
  float v1[3], v2[3], v3[3];
  cross3(s, s + 3, v1);
  cross3(s, s + 6, v2);
  cross3(s + 3, s + 6, v3);
  const float l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
  const float l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
  const float l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
  const float* v;
  float l;
  if (l1 >= l2 && l1 >= l3)
    v = v1, l = l1;
  else if (l2 >= l1 && l2 >= l3)
    v = v2, l = l2;
  else
    v = v3, l = l3;
  const float r = sqrtf(l);
  for (int a = 0; a < 3; ++a) vec[a] = v[a] / r;
}

/* normal and curvature from the points idx[0..k) of xyz */
void lo_point_normal(const float* xyz, const int32_t* idx, int k, const float* query, float nrm[3],
                     float* curvature) {
  if (k < 3) {
    nrm[0] = nrm[1] = nrm[2] = *curvature = NAN;
    return;
  }
  float accu[9] = {0};
  for (int s = 0; s < k; ++s) {
    const float* p = xyz + 3 * (size_t)idx[s];
    accu[0] += p[0] * p[0];
    accu[1] += p[0] * p[1];
    accu[2] += p[0] * p[2];
    accu[3] += p[1] * p[1];
    accu[4] += p[1] * p[2];
    accu[5] += p[2] * p[2];
    accu[6] += p[0];
    accu[7] += p[1];
    accu[8] += p[2];
  }
  for (int i = 0; i < 9; ++i) accu[i] /= (float)k;
  float cov[9];
  cov[0] = accu[0] - accu[6] * accu[6];
  cov[1] = cov[3] = accu[1] - accu[6] * accu[7];
  cov[2] = cov[6] = accu[2] - accu[6] * accu[8];
  cov[4] = accu[3] - accu[7] * accu[7];
  cov[5] = cov[7] = accu[4] - accu[7] * accu[8];
  cov[8] = accu[5] - accu[8] * accu[8];
  float ev;
  eigen33_smallest(cov, &ev, nrm);
  const float eig_sum = cov[0] + cov[4] + cov[8];
  *curvature = eig_sum != 0.f ? fabsf(ev / eig_sum) : 0.f;
  /* flipNormalTowardsViewpoint with the viewpoint at the origin */
  const float vx = 0.f - query[0], vy = 0.f - query[1], vz = 0.f - query[2];
  const float cos_theta = vx * nrm[0] + vy * nrm[1] + vz * nrm[2];
  if (cos_theta < 0.f) {
    nrm[0] *= -1.f;
    nrm[1] *= -1.f;
    nrm[2] *= -1.f;
  }
}

/* lamp_utils::ComputeNormals, search_method 0 (k nearest neighbours) */
void lo_compute_normals(const float* xyz, int n, int k, float* normals, float* curvature,
                        int threads) {
  if (n <= 0) return;
  if (k > n) k = n;
  lo_kdtree* t = lo_kdtree_build(xyz, n);
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)n * k);
  float* d2 = (float*)malloc(sizeof(float) * (size_t)n * k);
  lo_kdtree_knn(t, xyz, n, k, idx, d2, threads);
  for (int i = 0; i < n; ++i) {
    float c;
    lo_point_normal(xyz, idx + (size_t)i * k, k, xyz + 3 * (size_t)i, normals + 3 * (size_t)i, &c);
    if (curvature) curvature[i] = c;
  }
  free(idx);
  free(d2);
  lo_kdtree_free(t);
}

/* The normals pcl::HarrisKeypoint3D estimates for itself when none are handed in (the
 * lamp_utils::ComputeKeypoints overload without normals, lamp_utils/src/PointCloudUtils.cc:61-76):
 * pcl::NormalEstimation over a radius search with the Harris radius.  Neighbours = points with
 * squared distance < radius^2 (FLANN's strict test, the point itself included), summed in
 * ascending index (PCL visits them by ascending distance: last float bits may differ); fewer
 * than 3 neighbours give a NaN normal. */
void lo_compute_normals_radius(const float* xyz, int n, double radius, float* normals,
                               float* curvature) {
  if (n <= 0) return;
  const float r2 = (float)(radius * radius);
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  for (int i = 0; i < n; ++i) {
    const float* q = xyz + 3 * (size_t)i;
    int k = 0;
    if (isfinite(q[0]) && isfinite(q[1]) && isfinite(q[2]))
      for (int j = 0; j < n; ++j) {
        const float* p = xyz + 3 * (size_t)j;
        const float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        if (d2 < r2) idx[k++] = j;
      }
    float c;
    lo_point_normal(xyz, idx, k, q, normals + 3 * (size_t)i, &c);
    if (curvature) curvature[i] = c;
  }
  free(idx);
}

/* lamp_utils::ExtractNormals (PointCloudUtils.cc:17-35): the stored normals, unless the first
 * point's normal is all zero, in which case every normal is recomputed */
void lo_extract_normals(const float* xyz, const float* stored, int n, int k, float* normals,
                        int threads) {
  if (n <= 0) return;
  if (stored[0] == 0.f && stored[1] == 0.f && stored[2] == 0.f)
    lo_compute_normals(xyz, n, k, normals, NULL, threads);
  else
    memcpy(normals, stored, sizeof(float) * 3 * (size_t)n);
}
