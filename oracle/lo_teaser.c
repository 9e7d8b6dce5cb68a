/* lo_teaser.c — CPU restatement of the TEASER++ initial alignment that
 * IcpLoopComputation::GetTeaserInitialAlignment drives (loop_closure/src/IcpLoopComputation.cc:
 * 949-1081).  TEST INFRASTRUCTURE ONLY: the product library never links or calls this file.
 *
 * PARITY UNPINNED.  TEASER++ is a third-party dependency that is absent from the reference tree
 * (package.xml:25 `teaserpp`; install/lamp_http.rosinstall:10-12 names the branch
 * `feature/pcl-matcher` of MIT-SPARK/TEASER-plusplus, no commit pinned;
 * `Matcher::calculateKCorrespondences` and `SolutionEvaluator` exist only there), and the
 * reference's only test of this path is commented out (test_loop_computation.cc:101-123).
 * What follows restates the published algorithm (Yang, Shi, Carlone, "TEASER: Fast and
 * Certifiable Point Cloud Registration", and the upstream implementation's structure) with the
 * parameters of the call site: noise_bound 0.5, cbar2 1, no scale estimation, GNC-TLS rotation
 * (100 iterations, factor 1.4, cost threshold 1e-6), inlier selection PMC_HEU.  Parameters the
 * call site leaves alone keep the library's defaults; the one that matters is
 * `rotation_tim_graph = INLIER_GRAPH_FORMULATION::CHAIN` (it lives in the same Params struct as
 * the `inlier_selection_mode` the call site sets): the rotation is estimated from the TIMs of
 * consecutive clique members (ascending ids, the last one closing the chain to the first), not
 * from all pairs of them.
 *
 *   1. matches: every source descriptor with each of its k = 5 nearest target descriptors
 *   2. translation-invariant measurements (TIMs) of all match pairs i<j; a pair is
 *      scale-consistent when | |s_j - s_i| - |d_j - d_i| | <= 2 * noise_bound * sqrt(cbar2)
 *   3. greedy maximum-clique heuristic on the consistency graph (k-core ordered, no
 *      backtracking: the heuristic stage of the Parallel Maximum Clique library)
 *   4. GNC-TLS rotation on the chain of TIMs of the clique, weighted SVD alignment per iteration
 *   5. component-wise truncated-least-squares translation on the clique's matches
 *   6. the result replaces the odometry guess when its truncated error over the matches
 *      (sum min(d^2 / 100, 1)) is not larger than the guess's
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "lamp_oracle.h"

/* ---- 3x3 SVD by one-sided Jacobi, singular values descending, U and V proper up to sign */
static void svd3(const double H[9], double U[9], double S[3], double V[9]) {
  double A[9];
  memcpy(A, H, sizeof(A));
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    int rotated = 0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < 3; ++r) {
          alpha += A[r * 3 + p] * A[r * 3 + p];
          beta += A[r * 3 + q] * A[r * 3 + q];
          gamma += A[r * 3 + p] * A[r * 3 + q];
        }
        if (gamma == 0.0 || fabs(gamma) <= 1e-15 * sqrt(alpha * beta)) continue;
        rotated = 1;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int r = 0; r < 3; ++r) {
          const double ap = A[r * 3 + p], aq = A[r * 3 + q];
          A[r * 3 + p] = c * ap - s * aq;
          A[r * 3 + q] = s * ap + c * aq;
          const double vp = V[r * 3 + p], vq = V[r * 3 + q];
          V[r * 3 + p] = c * vp - s * vq;
          V[r * 3 + q] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  double sv[3];
  int o[3] = {0, 1, 2};
  for (int k = 0; k < 3; ++k)
    sv[k] = sqrt(A[k] * A[k] + A[3 + k] * A[3 + k] + A[6 + k] * A[6 + k]);
  for (int i = 0; i < 3; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (sv[o[j]] > sv[o[i]]) {
        const int t = o[i];
        o[i] = o[j];
        o[j] = t;
      }
  double Vs[9];
  for (int k = 0; k < 3; ++k) {
    S[k] = sv[o[k]];
    for (int r = 0; r < 3; ++r) {
      Vs[r * 3 + k] = V[r * 3 + o[k]];
      U[r * 3 + k] = S[k] > 0 ? A[r * 3 + o[k]] / S[k] : 0.0;
    }
  }
  memcpy(V, Vs, sizeof(Vs));
  /* complete U for vanishing singular values */
  const double tiny = 1e-14 * (S[0] > 0 ? S[0] : 1.0);
  if (S[0] <= 0) {
    for (int i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return;
  }
  if (S[1] <= tiny) { /* any unit vector orthogonal to u1 */
    const double u1[3] = {U[0], U[3], U[6]};
    int a = 0;
    if (fabs(u1[1]) < fabs(u1[a])) a = 1;
    if (fabs(u1[2]) < fabs(u1[a])) a = 2;
    double e[3] = {0, 0, 0};
    e[a] = 1.0;
    double u2[3] = {u1[1] * e[2] - u1[2] * e[1], u1[2] * e[0] - u1[0] * e[2],
                    u1[0] * e[1] - u1[1] * e[0]};
    const double n = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    for (int r = 0; r < 3; ++r) U[r * 3 + 1] = u2[r] / n;
  }
  if (S[2] <= tiny) {
    U[2] = U[3] * U[7] - U[6] * U[4];
    U[5] = U[6] * U[1] - U[0] * U[7];
    U[8] = U[0] * U[4] - U[3] * U[1];
  }
}

static double det3(const double M[9]) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
         M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* teaser::utils::svdRot: H = X diag(w) Y^T, R = V U^T with the last column of V flipped when
 * det(U) det(V) < 0 */
static void svd_rot(const double* X, const double* Y, const double* w, int n, double R[9]) {
  double H[9] = {0};
  for (int i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) H[a * 3 + b] += X[3 * i + a] * w[i] * Y[3 * i + b];
  double U[9], S[3], V[9];
  svd3(H, U, S, V);
  if (det3(U) * det3(V) < 0)
    for (int r = 0; r < 3; ++r) V[r * 3 + 2] = -V[r * 3 + 2];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b)
      R[a * 3 + b] = V[a * 3] * U[b * 3] + V[a * 3 + 1] * U[b * 3 + 1] + V[a * 3 + 2] * U[b * 3 + 2];
}

/* ---- 1. matches ------------------------------------------------------------------------- */
int lo_teaser_matches(const float* src_feat, int n_src, const float* tgt_feat, int n_tgt, int k,
                      int32_t* pairs) {
  if (k > n_tgt) k = n_tgt;
  if (k <= 0) return 0;
  float* nd = (float*)malloc(sizeof(float) * (size_t)k);
  int32_t* nn = (int32_t*)malloc(sizeof(int32_t) * (size_t)k);
  int count = 0;
  for (int i = 0; i < n_src; ++i) {
    for (int p = 0; p < k; ++p) nd[p] = INFINITY, nn[p] = -1;
    const float* f = src_feat + 33 * (size_t)i;
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
    for (int p = 0; p < k; ++p)
      if (nn[p] >= 0) {
        pairs[2 * count] = i;
        pairs[2 * count + 1] = nn[p];
        ++count;
      }
  }
  free(nd);
  free(nn);
  return count;
}

/* ---- 3. greedy clique ---------------------------------------------------------------------- */
static int adj_get(const uint8_t* adj, int n, int i, int j) { return adj[(size_t)i * n + j]; }

/* core number + 1 of every vertex (Batagelj-Zaversnik peeling) = bound on the size of a clique
 * through it */
static void core_bounds(const uint8_t* adj, int n, int* K) {
  int* deg = (int*)calloc((size_t)n, sizeof(int));
  char* gone = (char*)calloc((size_t)n, 1);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) deg[i] += adj_get(adj, n, i, j);
  int level = 0;
  for (int done = 0; done < n; ++done) {
    int v = -1;
    for (int i = 0; i < n; ++i) /* lowest remaining degree, lowest id */
      if (!gone[i] && (v < 0 || deg[i] < deg[v])) v = i;
    if (deg[v] > level) level = deg[v];
    K[v] = level + 1;
    gone[v] = 1;
    for (int j = 0; j < n; ++j)
      if (!gone[j] && adj_get(adj, n, v, j)) --deg[j];
  }
  free(deg);
  free(gone);
}

/* adj is a dense n x n 0/1 matrix.  Returns the clique size; clique holds ascending ids. */
int lo_teaser_clique(const uint8_t* adj, int n, int32_t* clique) {
  if (n <= 0) return 0;
  int* K = (int*)malloc(sizeof(int) * (size_t)n);
  core_bounds(adj, n, K);
  int* order = (int*)malloc(sizeof(int) * (size_t)n);
  for (int i = 0; i < n; ++i) order[i] = i;
  for (int i = 1; i < n; ++i) { /* stable: decreasing bound, ascending id */
    const int v = order[i];
    int p = i;
    while (p > 0 && K[order[p - 1]] < K[v]) {
      order[p] = order[p - 1];
      --p;
    }
    order[p] = v;
  }
  int* P = (int*)malloc(sizeof(int) * (size_t)n);
  int* C = (int*)malloc(sizeof(int) * (size_t)n);
  int mc = 0;
  for (int oi = 0; oi < n; ++oi) {
    const int v = order[oi];
    if (K[v] <= mc) continue;
    int np = 0;
    for (int j = 0; j < n; ++j)
      if (adj_get(adj, n, v, j) && K[j] > mc) P[np++] = j;
    if (np <= mc) continue;
    for (int i = 1; i < np; ++i) { /* stable: ascending bound */
      const int u = P[i];
      int p = i;
      while (p > 0 && K[P[p - 1]] > K[u]) {
        P[p] = P[p - 1];
        --p;
      }
      P[p] = u;
    }
    int nc = 0;
    while (np > 0) {
      const int u = P[--np];
      C[nc++] = u;
      int nr = 0;
      for (int i = 0; i < np; ++i)
        if (adj_get(adj, n, u, P[i]) && K[P[i]] > mc) P[nr++] = P[i];
      np = nr;
    }
    if (nc + 1 > mc) {
      mc = nc + 1;
      memcpy(clique, C, sizeof(int) * (size_t)nc);
      clique[nc] = v;
    }
  }
  for (int i = 1; i < mc; ++i) { /* ascending ids */
    const int v = clique[i];
    int p = i;
    while (p > 0 && clique[p - 1] > v) {
      clique[p] = clique[p - 1];
      --p;
    }
    clique[p] = v;
  }
  free(K);
  free(order);
  free(P);
  free(C);
  return mc;
}

/* ---- 4. GNC-TLS rotation ------------------------------------------------------------------- */
void lo_teaser_gnc_rotation(const double* src, const double* dst, int n, double noise_bound,
                            int max_iterations, double gnc_factor, double cost_threshold,
                            double R[9]) {
  for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  if (n <= 0) return;
  double* w = (double*)malloc(sizeof(double) * (size_t)n);
  double* res = (double*)malloc(sizeof(double) * (size_t)n);
  for (int i = 0; i < n; ++i) w[i] = 1.0;
  double mu = 1.0, prev_cost = INFINITY;
  double nb2 = noise_bound * noise_bound;
  if (nb2 < 1e-16) nb2 = 1e-2;
  for (int it = 0; it < max_iterations; ++it) {
    svd_rot(src, dst, w, n, R);
    double max_res = 0;
    for (int i = 0; i < n; ++i) {
      double r2 = 0;
      for (int a = 0; a < 3; ++a) {
        const double d = dst[3 * i + a] - (R[a * 3] * src[3 * i] + R[a * 3 + 1] * src[3 * i + 1] +
                                           R[a * 3 + 2] * src[3 * i + 2]);
        r2 += d * d;
      }
      res[i] = r2;
      if (r2 > max_res) max_res = r2;
    }
    if (it == 0) {
      mu = 1.0 / (2.0 * max_res / nb2 - 1.0);
      if (mu <= 0) break; /* every residual already inside the noise bound */
    }
    const double th1 = (mu + 1) / mu * nb2, th2 = mu / (mu + 1) * nb2;
    double cost = 0;
    for (int i = 0; i < n; ++i) {
      cost += w[i] * res[i];
      if (res[i] >= th1)
        w[i] = 0;
      else if (res[i] <= th2)
        w[i] = 1;
      else
        w[i] = sqrt(nb2 * mu * (mu + 1) / res[i]) - mu;
    }
    const double cost_diff = fabs(cost - prev_cost);
    mu *= gnc_factor;
    prev_cost = cost;
    if (cost_diff < cost_threshold) break;
  }
  free(w);
  free(res);
}

/* ---- 5. scalar truncated least squares (adaptive voting) ----------------------------------- */
typedef struct {
  double v;
  int id, seq;
} tls_center;
static int cmp_center(const void* a, const void* b) {
  const tls_center* x = (const tls_center*)a;
  const tls_center* y = (const tls_center*)b;
  if (x->v < y->v) return -1;
  if (x->v > y->v) return 1;
  return x->seq - y->seq;
}

double lo_teaser_tls_scalar(const double* X, int n, double range) {
  tls_center* h = (tls_center*)malloc(sizeof(tls_center) * 2 * (size_t)n);
  for (int i = 0; i < n; ++i) {
    h[2 * i] = (tls_center){X[i] - range, i + 1, 2 * i};
    h[2 * i + 1] = (tls_center){X[i] + range, -i - 1, 2 * i + 1};
  }
  qsort(h, 2 * (size_t)n, sizeof(tls_center), cmp_center);
  const double weight = 1.0 / (range * range);
  double ranges_sum = range * n, dot_xw = 0, dot_w = 0, sum_x = 0, sum_xx = 0;
  int card = 0, have = 0;
  double best_cost = 0, best_x = 0;
  for (int i = 0; i < 2 * n; ++i) {
    const int idx = abs(h[i].id) - 1;
    const int eps = h[i].id > 0 ? 1 : -1;
    card += eps;
    dot_w += eps * weight;
    dot_xw += eps * weight * X[idx];
    ranges_sum -= eps * range;
    sum_x += eps * X[idx];
    sum_xx += eps * X[idx] * X[idx];
    if (card <= 0) continue; /* empty consensus set: 0/0 in the closed form */
    const double x_hat = dot_xw / dot_w;
    const double cost = (card * x_hat * x_hat + sum_xx - 2 * sum_x * x_hat) + ranges_sum;
    if (!have || cost < best_cost) have = 1, best_cost = cost, best_x = x_hat;
  }
  free(h);
  return best_x;
}

/* ---- 2-5: the solver ----------------------------------------------------------------------- */
int lo_teaser_solve(const double* src, const double* dst, int n, double noise_bound, double cbar2,
                    int rotation_max_iterations, double gnc_factor, double cost_threshold,
                    double R[9], double t[3], int32_t* clique_out, int* n_clique_out) {
  for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  t[0] = t[1] = t[2] = 0;
  *n_clique_out = 0;
  if (n <= 0) return 0;
  const double beta = 2.0 * noise_bound * sqrt(cbar2);
  uint8_t* adj = (uint8_t*)calloc((size_t)n * n, 1);
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      double a = 0, b = 0;
      for (int c = 0; c < 3; ++c) {
        const double ds = src[3 * j + c] - src[3 * i + c], dd = dst[3 * j + c] - dst[3 * i + c];
        a += ds * ds;
        b += dd * dd;
      }
      if (fabs(sqrt(a) - sqrt(b)) <= beta) adj[(size_t)i * n + j] = adj[(size_t)j * n + i] = 1;
    }
  const int m = lo_teaser_clique(adj, n, clique_out);
  free(adj);
  *n_clique_out = m;
  if (m <= 1) return 0; /* the solver reports an invalid solution: identity */
  /* chain graph of the clique: member i -> member i + 1, the last -> the first */
  const size_t M = (size_t)m;
  double* st = (double*)malloc(sizeof(double) * 3 * M);
  double* dt = (double*)malloc(sizeof(double) * 3 * M);
  for (int i = 0; i < m; ++i) {
    const int root = clique_out[i], leaf = clique_out[i + 1 < m ? i + 1 : 0];
    for (int c = 0; c < 3; ++c) {
      st[3 * (size_t)i + c] = src[3 * leaf + c] - src[3 * root + c];
      dt[3 * (size_t)i + c] = dst[3 * leaf + c] - dst[3 * root + c];
    }
  }
  lo_teaser_gnc_rotation(st, dt, (int)M, noise_bound, rotation_max_iterations, gnc_factor,
                         cost_threshold, R);
  free(st);
  free(dt);
  double* v = (double*)malloc(sizeof(double) * (size_t)m);
  for (int a = 0; a < 3; ++a) {
    for (int i = 0; i < m; ++i) {
      const double* s = src + 3 * clique_out[i];
      v[i] = dst[3 * clique_out[i] + a] - (R[a * 3] * s[0] + R[a * 3 + 1] * s[1] + R[a * 3 + 2] * s[2]);
    }
    t[a] = lo_teaser_tls_scalar(v, m, cbar2 * noise_bound);
  }
  free(v);
  return 1;
}

/* ---- 6. truncated error of a transform over the matches ------------------------------------- */
double lo_teaser_error_metric(const double* src, const double* dst, int n, const double R[9],
                              const double t[3], double threshold) {
  double err = 0;
  for (int i = 0; i < n; ++i) {
    double d2 = 0;
    for (int a = 0; a < 3; ++a) {
      const double d = (R[a * 3] * src[3 * i] + R[a * 3 + 1] * src[3 * i + 1] +
                        R[a * 3 + 2] * src[3 * i + 2] + t[a]) - dst[3 * i + a];
      d2 += d * d;
    }
    err += d2 <= threshold ? d2 / threshold : 1.0;
  }
  return err;
}

/* GetTeaserInitialAlignment from the keypoints on.  T is the guess on entry and the initial
 * alignment on return.  Returns 1 when the TEASER++ transform replaced the guess. */
int lo_teaser_initial_alignment(const float* src_kp, const float* src_feat, int n_src,
                                const float* tgt_kp, const float* tgt_feat, int n_tgt, float T[16],
                                int* n_matches_out, int* n_clique_out) {
  if (n_matches_out) *n_matches_out = 0;
  if (n_clique_out) *n_clique_out = 0;
  if (n_src == 0 || n_tgt == 0) return 0; /* IcpLoopComputation.cc:990-992 */
  int32_t* pairs = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)n_src * 5);
  const int n = lo_teaser_matches(src_feat, n_src, tgt_feat, n_tgt, 5, pairs);
  if (n_matches_out) *n_matches_out = n;
  int used = 0;
  if (n > 10) { /* IcpLoopComputation.cc:1002 */
    double* s = (double*)malloc(sizeof(double) * 3 * (size_t)n);
    double* d = (double*)malloc(sizeof(double) * 3 * (size_t)n);
    for (int i = 0; i < n; ++i)
      for (int c = 0; c < 3; ++c) {
        s[3 * i + c] = src_kp[3 * pairs[2 * i] + c];
        d[3 * i + c] = tgt_kp[3 * pairs[2 * i + 1] + c];
      }
    int32_t* clique = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    int m = 0;
    double R[9], t[3], Ro[9], to[3];
    lo_teaser_solve(s, d, n, 0.5, 1.0, 100, 1.4, 1e-6, R, t, clique, &m);
    if (n_clique_out) *n_clique_out = m;
    for (int a = 0; a < 3; ++a) {
      for (int b = 0; b < 3; ++b) Ro[a * 3 + b] = (double)T[a * 4 + b];
      to[a] = (double)T[a * 4 + 3];
    }
    const double e_teaser = lo_teaser_error_metric(s, d, n, R, t, 100.0);
    const double e_odom = lo_teaser_error_metric(s, d, n, Ro, to, 100.0);
    if (e_teaser <= e_odom) { /* IcpLoopComputation.cc:1065-1068 */
      for (int i = 0; i < 16; ++i) T[i] = (i == 15) ? 1.f : 0.f;
      for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) T[a * 4 + b] = (float)R[a * 3 + b];
        T[a * 4 + 3] = (float)t[a];
      }
      used = 1;
    }
    free(s);
    free(d);
    free(clique);
  }
  free(pairs);
  return used;
}
