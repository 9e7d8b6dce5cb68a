/*
 * lo_icp_loop.c — CPU restatement of IcpLoopComputation::PerformAlignment and the
 * PointCloudUtils helpers it calls (TEST INFRASTRUCTURE ONLY, see lamp_oracle.h).
 *
 * Follows loop_closure/src/IcpLoopComputation.cc and lamp_utils/src/PointCloudUtils.cc of
 * the reference; geometry_utils (Rot3::Roll/Pitch/Yaw, PoseInverse) and GTSAM (Pose3) are
 * not under /root/reference and are restated from their published formulas.
 */
#include "lamp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* NormalizePCloud (PointCloudUtils.cc:39-59): pcl::compute3DCentroid accumulates in float
 * in point order; the mean distance is a float running sum; the scaling transform is applied
 * with pcl::transformPointCloud's arithmetic. */
void lo_normalize_cloud(const float* xyz, int n, float* out) {
  float c[3] = {0.f, 0.f, 0.f};
  for (int i = 0; i < n; ++i) {
    c[0] += xyz[3 * i];
    c[1] += xyz[3 * i + 1];
    c[2] += xyz[3 * i + 2];
  }
  c[0] /= (float)n;
  c[1] /= (float)n;
  c[2] /= (float)n;
  float dist = 0.f;
  for (int i = 0; i < n; ++i) {
    const float dx = xyz[3 * i] - c[0], dy = xyz[3 * i + 1] - c[1], dz = xyz[3 * i + 2] - c[2];
    dist = dist + sqrtf(dx * dx + (dy * dy + dz * dz)); /* Eigen redux of a Vector3f: x2 + (y2 + z2) */
  }
  const float factor = (float)(size_t)n / dist;
  float T[16] = {0};
  T[0] = factor;
  T[5] = factor;
  T[10] = factor;
  T[3] = -factor * c[0];
  T[7] = -factor * c[1];
  T[11] = -factor * c[2];
  T[15] = -factor * 1.0f; /* block(0,3,4,1) = -factor * centroid_4d, unused by the transform */
  lo_transform_points_f32(T, xyz, n, out);
}

/* ComputeAp_ForPoint2PlaneICP (PointCloudUtils.cc:115-163) */
void lo_compute_ap(const float* query_normalized, int n_query, const float* ref_normals,
                   int n_ref_normals, const int64_t* corr, int n_corr, const float T[16],
                   double Ap[36]) {
  for (int i = 0; i < 36; ++i) Ap[i] = 0.0;
  double R[9];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) R[a * 3 + b] = (double)T[a * 4 + b];
  for (int i = 0; i < n_query; ++i) {
    if (i >= n_corr) continue;
    const double a_i[3] = {query_normalized[3 * i], query_normalized[3 * i + 1],
                           query_normalized[3 * i + 2]};
    double n_i[3] = {0, 0, 0};
    if (ref_normals && corr[i] >= 0 && (int64_t)n_ref_normals > corr[i]) {
      n_i[0] = ref_normals[3 * corr[i]];
      n_i[1] = ref_normals[3 * corr[i] + 1];
      n_i[2] = ref_normals[3 * corr[i] + 2];
    }
    if (isnan(a_i[0]) || isnan(a_i[1]) || isnan(a_i[2]) || isnan(n_i[0]) || isnan(n_i[1]) ||
        isnan(n_i[2]))
      continue;
    const double Rn[3] = {R[0] * n_i[0] + R[1] * n_i[1] + R[2] * n_i[2],
                          R[3] * n_i[0] + R[4] * n_i[1] + R[5] * n_i[2],
                          R[6] * n_i[0] + R[7] * n_i[1] + R[8] * n_i[2]};
    const double H[6] = {a_i[1] * Rn[2] - a_i[2] * Rn[1], a_i[2] * Rn[0] - a_i[0] * Rn[2],
                         a_i[0] * Rn[1] - a_i[1] * Rn[0], Rn[0], Rn[1], Rn[2]};
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) Ap[r * 6 + c] += H[r] * H[c];
  }
}

/* ComputeIcpObservability (PointCloudUtils.cc:165-196) with normals already present in the
 * cloud (ExtractNormals copy branch, PointCloudUtils.cc:28-34) */
void lo_icp_observability(const float* xyz, const float* normals, int n, double eig[3]) {
  float* nrm = (float*)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
  int64_t* c = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  lo_normalize_cloud(xyz, n, nrm);
  for (int i = 0; i < n; ++i) c[i] = i;
  const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  double Ap[36];
  lo_compute_ap(nrm, n, normals, n, c, n, I, Ap);
  double B[9];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) B[a * 3 + b] = Ap[(3 + a) * 6 + 3 + b];
  lo_sym_eig(B, 3, eig, NULL);
  free(nrm);
  free(c);
}

static void fixed_cov(double rot_sigma, double trans_sigma, double cov[36]) {
  for (int i = 0; i < 36; ++i) cov[i] = 0.0;
  for (int i = 0; i < 3; ++i) cov[i * 6 + i] = rot_sigma * rot_sigma;
  for (int i = 3; i < 6; ++i) cov[i * 6 + i] = trans_sigma * trans_sigma;
}

/* ComputeICPCovariancePointPlane (IcpLoopComputation.cc:707-768).  The reference assigns
 * only the diagonal in the fixed-covariance branches and leaves the caller's other entries
 * untouched; the oracle zeroes them. */
void lo_icp_cov_point_plane(const float* scan1_xyz, int n1, const float* scan2_normals, int n2,
                            const int64_t* corr, int n_corr, const float T[16], double rot_sigma,
                            double trans_sigma, double cov[36], double Ap_out[36]) {
  float* qn = (float*)malloc(sizeof(float) * 3 * (size_t)(n1 > 0 ? n1 : 1));
  lo_normalize_cloud(scan1_xyz, n1, qn);
  double Ap[36];
  lo_compute_ap(qn, n1, scan2_normals, n2, corr, n_corr, T, Ap);
  free(qn);
  if (Ap_out) memcpy(Ap_out, Ap, sizeof(Ap));
  if (lo_mat6_det(Ap) == 0) {
    fixed_cov(rot_sigma, trans_sigma, cov);
    return;
  }
  double inv[36];
  lo_mat6_inverse(Ap, inv);
  for (int i = 0; i < 36; ++i) cov[i] = 0.05 * 0.05 * inv[i];
  double L[36], D[6];
  int tr[6];
  lo_ldlt(cov, L, D, tr);
  const double lower_bound = 1e-12, upper_bound = 1;
  int recompute = 0;
  for (int i = 0; i < 6; ++i) {
    if (D[i] <= 0) {
      D[i] = lower_bound;
      recompute = 1;
    }
    if (D[i] > upper_bound) {
      D[i] = upper_bound;
      recompute = 1;
    }
  }
  if (recompute) { /* L * D * L^T, transpositions ignored as in the reference (:757) */
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        double s = 0;
        for (int k = 0; k < 6; ++k) s += L[i * 6 + k] * D[k] * L[j * 6 + k];
        cov[i * 6 + j] = s;
      }
  }
  for (int i = 0; i < 36; ++i)
    if (isnan(cov[i])) {
      fixed_cov(rot_sigma, trans_sigma, cov);
      break;
    }
}

/* ComputeICPCovariancePointPoint (IcpLoopComputation.cc:770-908).  gu::Rot3::Roll/Pitch/Yaw
 * = ZYX Euler angles of T's rotation. */
int lo_icp_cov_point_point(const float* aligned_xyz, int n, const float T[16], double fitness,
                           double cov[36]) {
  const double t_x = T[3], t_y = T[7], t_z = T[11];
  const double R20 = T[8], R21 = T[9], R22 = T[10], R10 = T[4], R00 = T[0], R12 = T[6],
               R11 = T[5];
  double r, p, y;
  p = -asin(R20);
  if (cos(p) > 1e-6) {
    r = atan2(R21, R22);
    y = atan2(R10, R00);
  } else {
    r = atan2(-R12, R11);
    y = 0.0;
  }
  double H[36] = {0};
  const double cr = cos(r), sr = sin(r), cp = cos(p), sp = sin(p), cy = cos(y), sy = sin(y);
  for (int i = 0; i < n; ++i) {
    const double p_x = aligned_xyz[3 * i], p_y = aligned_xyz[3 * i + 1],
                 p_z = aligned_xyz[3 * i + 2];
    double J[18] = {0};
    const double ex = (t_x - p_x + p_z * cp - p_x * cy * sp + p_y * sp * sy);
    J[0 * 6 + 1] = -2.0 * (p_z * sp + p_x * cp * cy - p_y * cp * sy) * ex;
    J[0 * 6 + 2] = 2.0 * (p_y * cy * sp + p_x * sp * sy) * ex;
    J[0 * 6 + 3] = 2.0 * t_x - 2.0 * p_x + 2.0 * p_z * cp - 2.0 * p_x * cy * sp + 2.0 * p_y * sp * sy;
    const double ey = (p_y - t_y + p_x * (sr * sy - cp * cr * cy) + p_y * (cy * sr + cp * cr * sy) -
                       p_z * cr * sp);
    J[1 * 6 + 0] =
        2.0 * (p_x * (cr * sy + cp * cy * sr) + p_y * (cr * cy - cp * sr * sy) + p_z * sp * sr) * ey;
    J[1 * 6 + 1] = -2.0 * (p_z * cp * cr - p_x * cr * cy * sp + p_y * cr * sp * sy) * ey;
    J[1 * 6 + 2] =
        2.0 * (p_x * (cy * sr + cp * cr * sy) - p_y * (sr * sy - cp * cr * cy)) * ey;
    J[1 * 6 + 4] = 2.0 * t_y - 2.0 * p_y - 2.0 * p_x * (sr * sy - cp * cr * cy) -
                   2.0 * p_y * (cy * sr + cp * cr * sy) + 2.0 * p_z * cr * sp;
    const double ez = (t_z - p_z + p_x * (cr * sy + cp * cy * sr) + p_y * (cr * cy - cp * sr * sy) +
                       p_z * sp * sr);
    J[2 * 6 + 0] =
        -2.0 * (p_x * (sr * sy - cp * cr * cy) + p_y * (cy * sr + cp * cr * sy) - p_z * cr * sp) * ez;
    J[2 * 6 + 1] = 2.0 * (p_z * cp * sr - p_x * cy * sp * sr + p_y * sp * sr * sy) * ez;
    J[2 * 6 + 2] =
        2.0 * (p_x * (cr * cy - cp * sr * sy) - p_y * (cr * sy + cp * cy * sr)) * ez;
    J[2 * 6 + 5] = 2.0 * t_z - 2.0 * p_z + 2.0 * p_x * (cr * sy + cp * cy * sr) +
                   2.0 * p_y * (cr * cy - cp * sr * sy) + 2.0 * p_z * sp * sr;
    for (int a = 0; a < 6; ++a)
      for (int b = 0; b < 6; ++b) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += J[k * 6 + a] * J[k * 6 + b];
        H[a * 6 + b] += s;
      }
  }
  double inv[36];
  lo_mat6_inverse(H, inv);
  for (int i = 0; i < 36; ++i) cov[i] = inv[i] * fitness;
  for (int i = 0; i < 36; ++i)
    if (!isfinite(cov[i])) {
      for (int k = 0; k < 36; ++k) cov[k] = (k % 7 == 0) ? 1000.0 : 0.0;
      return 0;
    }
  double ev[6], V[36];
  lo_sym_eig(cov, 6, ev, V);
  for (int i = 0; i < 6; ++i) {
    if (ev[i] < 0.001) ev[i] = 0.001;
    if (ev[i] > 1000) ev[i] = 1000;
  }
  for (int a = 0; a < 6; ++a)
    for (int b = 0; b < 6; ++b) {
      double s = 0;
      for (int k = 0; k < 6; ++k) s += V[a * 6 + k] * ev[k] * V[b * 6 + k];
      cov[a * 6 + b] = s;
    }
  return 1;
}

/* IcpLoopComputation::PerformAlignment (IcpLoopComputation.cc:420-648) */
int lo_perform_alignment(const lo_icp_params* p, const float* src_xyz, const float* src_normals,
                         int n_src, const float* tgt_xyz, const float* tgt_normals, int n_tgt,
                         const float* scan1_xyz, int n_scan1, const float* scan2_normals,
                         int n_scan2, const lo_pose* pose_21_odom, const lo_pose* pose_21_cand,
                         lo_alignment* out) {
  memset(out, 0, sizeof(*out));
  if (n_src == 0 || n_tgt == 0) { /* :458-465 */
    out->status = LO_FAIL_EMPTY;
    return out->status;
  }
  lo_gicp_params gp = p->gicp;
  if (n_src < 20) gp.k_correspondences = n_src; /* :487-489 */

  /* initial guess (:495-543) */
  float guess[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const lo_pose* src_pose = NULL;
  switch (p->init_method) {
  case LO_INIT_IDENTITY: break;
  case LO_INIT_ODOMETRY: src_pose = pose_21_odom; break;
  case LO_INIT_ODOM_ROTATION: src_pose = pose_21_odom; break;
  case LO_INIT_CANDIDATE: src_pose = pose_21_cand; break;
  case LO_INIT_FEATURES:
  case LO_INIT_TEASERPP:
    out->status = LO_FAIL_UNSUPPORTED_INIT;
    return out->status;
  default: break;
  }
  if (src_pose) {
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) guess[a * 4 + b] = (float)src_pose->R[a * 3 + b];
    if (p->init_method != LO_INIT_ODOM_ROTATION)
      for (int a = 0; a < 3; ++a) guess[a * 4 + 3] = (float)src_pose->t[a];
  }

  lo_gicp_result gr;
  lo_gicp_align(&gp, src_xyz, src_normals, n_src, tgt_xyz, tgt_normals, n_tgt, guess, &gr, NULL,
                NULL, 0);
  memcpy(out->T_icp, gr.final_transformation, sizeof(out->T_icp));
  out->iterations = gr.iterations;
  out->converged = gr.converged;
  const float* T = out->T_icp;

  /* correspondences (:553-566) + fitness (:586) share the same 1-NN pass */
  int32_t* corr32 = (int32_t*)malloc(sizeof(int32_t) * (size_t)n_src);
  const double fitness =
      lo_fitness_score(src_xyz, n_src, tgt_xyz, n_tgt, T, corr32, gp.use_kdtree, gp.threads);
  if (p->covariance_method == LO_COV_POINT2PLANE) {
    /* "Catch nan of infs in icp result" (:556-558): icp_result is T applied to the source, so a
     * non-finite T spoils every point and a non-finite source point stays what it is */
    int bad = 0;
    for (int i = 0; i < 16; ++i) bad |= !isfinite(T[i]);
    for (int i = 0; i < n_src && !bad; ++i)
      bad = !isfinite(src_xyz[3 * i]) || !isfinite(src_xyz[3 * i + 1]) || !isfinite(src_xyz[3 * i + 2]);
    if (bad) {
      free(corr32);
      out->status = LO_FAIL_NONFINITE;
      return out->status;
    }
  }
  lo_pose delta;
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) delta.R[a * 3 + b] = (double)T[a * 4 + b];
    delta.t[a] = (double)T[a * 4 + 3];
  }
  if (!gr.converged) { /* :580-584 */
    free(corr32);
    out->status = LO_FAIL_NOT_CONVERGED;
    return out->status;
  }
  out->fitness = fitness;
  if (fitness > p->max_tolerable_fitness) { /* :588-593 */
    free(corr32);
    out->status = LO_FAIL_FITNESS;
    return out->status;
  }
  { /* :597-614 */
    lo_pose diff;
    double xi[6];
    lo_pose_between(pose_21_odom, &delta, &diff);
    lo_pose_logmap(&diff, xi);
    const double trans_diff = sqrt(xi[3] * xi[3] + xi[4] * xi[4] + xi[5] * xi[5]);
    const double rot_diff = sqrt(xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2]);
    if (p->transform_thresholding &&
        (trans_diff > p->max_translation || rot_diff > p->max_rotation_deg * M_PI / 180.0)) {
      free(corr32);
      out->status = LO_FAIL_TRANSFORM;
      return out->status;
    }
  }
  lo_pose_inverse(&delta, &out->delta); /* :617 */
  if (p->use_fixed_covariances) {
    fixed_cov(p->rot_sigma, p->trans_sigma, out->covariance);
  } else if (p->covariance_method == LO_COV_POINT2POINT) {
    float* aligned = (float*)malloc(sizeof(float) * 3 * (size_t)n_src);
    lo_transform_points_f32(T, src_xyz, n_src, aligned);
    lo_icp_cov_point_point(aligned, n_src, T, fitness, out->covariance);
    free(aligned);
  } else {
    int64_t* corr = (int64_t*)malloc(sizeof(int64_t) * (size_t)n_src);
    for (int i = 0; i < n_src; ++i) corr[i] = corr32[i];
    lo_icp_cov_point_plane(scan1_xyz, n_scan1, scan2_normals, n_scan2, corr, n_src, T,
                           p->rot_sigma, p->trans_sigma, out->covariance, NULL);
    free(corr);
  }
  free(corr32);
  out->status = LO_OK;
  return out->status;
}
