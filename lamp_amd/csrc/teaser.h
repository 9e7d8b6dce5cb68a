// teaser.h — host side of the TEASERPP initialisation
// (IcpLoopComputation::GetTeaserInitialAlignment, loop_closure/src/IcpLoopComputation.cc:949-1081
// -> teaser::RobustRegistrationSolver with noise_bound 0.5, cbar2 1, GNC-TLS rotation, PMC_HEU).
//
// Split of the work: descriptor k-NN matches and the O(n^2) scale-consistency graph of the
// matches are built on the GPU (feature_init.hip) and arrive here as a bit matrix; what is left is
// graph and small dense algebra on the inlier clique: core numbers, the greedy clique heuristic
// (both also on the GPU, teaser_graph.hip), graduated non-convexity for the rotation (a weighted
// 3x3 SVD per iteration over the chain of differences between consecutive clique members), and a
// per-axis truncated-least-squares vote for the translation.
// TEASER++ itself is not part of the reference tree; the algorithm follows its publication.
#pragma once

#include <algorithm>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace lgicp {
namespace teaser {

struct BitGraph {
  int n = 0, words = 0;
  std::vector<uint64_t> bits;  // [n][words], symmetric, zero diagonal
  const uint64_t* row(int i) const { return bits.data() + (size_t)i * words; }
  bool has(int i, int j) const { return (row(i)[j >> 6] >> (j & 63)) & 1; }
};

// core number + 1 per vertex: an upper bound on the size of a clique through it
inline std::vector<int> core_bounds(const BitGraph& g) {
  const int n = g.n;
  std::vector<int> deg(n, 0), K(n, 0);
  int max_deg = 0;
  for (int i = 0; i < n; ++i) {
    int d = 0;
    for (int w = 0; w < g.words; ++w) d += __builtin_popcountll(g.row(i)[w]);
    deg[i] = d;
    max_deg = std::max(max_deg, d);
  }
  // bucket peeling (Batagelj-Zaversnik)
  std::vector<int> bin(max_deg + 2, 0), pos(n), vert(n);
  for (int i = 0; i < n; ++i) bin[deg[i]]++;
  int start = 0;
  for (int d = 0; d <= max_deg; ++d) {
    const int c = bin[d];
    bin[d] = start;
    start += c;
  }
  for (int i = 0; i < n; ++i) {
    pos[i] = bin[deg[i]]++;
    vert[pos[i]] = i;
  }
  for (int d = max_deg; d > 0; --d) bin[d] = bin[d - 1];
  bin[0] = 0;
  for (int i = 0; i < n; ++i) {
    const int v = vert[i];
    K[v] = deg[v] + 1;
    const uint64_t* r = g.row(v);
    for (int w = 0; w < g.words; ++w) {
      uint64_t m = r[w];
      while (m) {
        const int u = (w << 6) + __builtin_ctzll(m);
        m &= m - 1;
        if (deg[u] > deg[v]) {
          const int du = deg[u], pu = pos[u], pw = bin[du], wv = vert[pw];
          if (u != wv) {
            pos[u] = pw, vert[pu] = wv;
            pos[wv] = pu, vert[pw] = u;
          }
          bin[du]++;
          deg[u]--;
        }
      }
    }
  }
  return K;
}

// Greedy clique heuristic: vertices in order of decreasing bound (ascending id among equals);
// from each, repeatedly move to the common neighbour with the largest (bound, id).  A start
// only looks at vertices whose bound exceeds the best size so far, so what it finds depends on
// that size; the definition is the serial one: starts in order, each seeing the best size its
// predecessors left.  With threads > 1 a window of starts is evaluated side by side against the
// size at the window's beginning; as long as none of them improves it - the rule once the first
// few starts have found a large clique - their results are exactly the serial ones, and when one
// does, the starts after it in the window are repeated with the new size.  Same clique, bit for
// bit, for any number of threads.
inline std::vector<int> greedy_clique(const BitGraph& g, int threads = 1) {
  const int n = g.n;
  std::vector<int> best;
  if (n <= 0) return best;
  const std::vector<int> K = core_bounds(g);
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return K[a] > K[b]; });
  // candidates only ever leave the set, so the preferred one (largest bound, then largest id)
  // is found by walking a list sorted that way instead of scanning the whole set every time
  std::vector<int> pref(n);
  for (int i = 0; i < n; ++i) pref[i] = i;
  std::sort(pref.begin(), pref.end(), [&](int a, int b) { return K[a] != K[b] ? K[a] > K[b] : a > b; });
  // one start: the clique found from v (without v) when the best size so far is mc; empty when
  // v has too few eligible neighbours to beat it.  A pure function of its arguments.
  auto start_from = [&](int v, int mc, std::vector<uint64_t>& cand, std::vector<int>& C) {
    C.clear();
    int np = 0;
    for (int w = 0; w < g.words; ++w) {
      uint64_t m = g.row(v)[w], keep = 0;
      while (m) {
        const int b = __builtin_ctzll(m);
        m &= m - 1;
        if (K[(w << 6) + b] > mc) keep |= 1ull << b, ++np;
      }
      cand[w] = keep;
    }
    if (np <= mc) return;
    for (int at = 0; at < n; ++at) {
      const int u = pref[at];
      if (!((cand[u >> 6] >> (u & 63)) & 1)) continue;
      C.push_back(u);
      const uint64_t* r = g.row(u);
      int left = 0;
      for (int w = 0; w < g.words; ++w) {
        cand[w] &= r[w];
        left += __builtin_popcountll(cand[w]);
      }
      // v, the vertices taken so far and every remaining candidate together cannot beat mc:
      // whatever this start ends with would be discarded
      if (1 + (int)C.size() + left <= mc) return;
    }
  };
  int mc = 0;
  auto apply = [&](int v, const std::vector<int>& C) {
    if (C.empty() || (int)C.size() + 1 <= mc) return false;
    mc = (int)C.size() + 1;
    best = C;
    best.push_back(v);
    return true;
  };
  // helpers for this call: at most threads - 1, and at most 16 over all calls running at once
  // (several pairs of a submit are solved side by side)
  static std::atomic<int> helpers_out{0};
  int n_helpers = 0;
  if (threads > 1) {
    const int want = threads - 1;
    const int before = helpers_out.fetch_add(want);
    n_helpers = std::max(0, std::min(want, 16 - before));
    helpers_out.fetch_sub(want - n_helpers);
  }
  const size_t cap = (size_t)(n_helpers + 1) * (n_helpers ? 16 : 1);  // starts per window
  std::vector<int> window;
  std::vector<std::vector<int>> found(cap);
  std::vector<uint64_t> cand0((size_t)g.words);
  // Window state shared with the helpers.  A helper may only work on a window between the moment
  // the main thread opens it and the moment it closes it again (both under the mutex, the latter
  // only once no helper is inside): the window's vectors and counters are rewritten in between.
  std::mutex mu;
  std::condition_variable cv, cv_done;
  unsigned long generation = 0;
  bool stop = false, open = false;
  int inside = 0;
  int mc0 = 0;
  std::atomic<size_t> next{0}, done{0};
  auto drain = [&](std::vector<uint64_t>& cand) {
    for (;;) {
      const size_t k = next.fetch_add(1);
      if (k >= window.size()) return;
      start_from(window[k], mc0, cand, found[k]);
      done.fetch_add(1);
    }
  };
  std::vector<std::thread> pool;
  for (int h = 0; h < n_helpers; ++h)
    pool.emplace_back([&] {
      std::vector<uint64_t> cand((size_t)g.words);
      unsigned long seen = 0;
      for (;;) {
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return stop || (open && generation != seen); });
          if (stop) return;
          seen = generation;
          ++inside;
        }
        drain(cand);
        {
          std::lock_guard<std::mutex> lk(mu);
          --inside;
        }
        cv_done.notify_all();
      }
    });
  int at = 0;
  while (at < n) {
    window.clear();
    while (at < n && window.size() < cap) {
      const int v = order[at++];
      if (K[v] > mc) window.push_back(v);
    }
    if (window.empty()) break;
    mc0 = mc;
    next.store(0), done.store(0);
    if (n_helpers) {
      {
        std::lock_guard<std::mutex> lk(mu);
        ++generation;
        open = true;
      }
      cv.notify_all();
    }
    drain(cand0);
    if (n_helpers) {  // sleep, do not spin: the machine's cores are shared with the other pairs' threads
      std::unique_lock<std::mutex> lk(mu);
      cv_done.wait(lk, [&] { return inside == 0 && done.load() >= window.size(); });
      open = false;
    }
    for (size_t k = 0; k < window.size(); ++k) {
      if (mc != mc0) {  // an earlier start of this window raised the size: this one saw a stale one
        if (K[window[k]] <= mc) continue;
        start_from(window[k], mc, cand0, found[k]);
      }
      apply(window[k], found[k]);
    }
  }
  if (n_helpers) {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    for (std::thread& t : pool) t.join();
    helpers_out.fetch_sub(n_helpers);
  }
  std::sort(best.begin(), best.end());
  return best;
}

// ---- 3x3 SVD (one-sided Jacobi), singular values descending
inline void svd3(const double H[9], double U[9], double S[3], double V[9]) {
  double A[9];
  std::memcpy(A, H, sizeof(A));
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < 3; ++r) {
          alpha += A[r * 3 + p] * A[r * 3 + p];
          beta += A[r * 3 + q] * A[r * 3 + q];
          gamma += A[r * 3 + p] * A[r * 3 + q];
        }
        if (gamma == 0.0 || std::fabs(gamma) <= 1e-15 * std::sqrt(alpha * beta)) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
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
  for (int k = 0; k < 3; ++k) sv[k] = std::sqrt(A[k] * A[k] + A[3 + k] * A[3 + k] + A[6 + k] * A[6 + k]);
  for (int i = 0; i < 3; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (sv[o[j]] > sv[o[i]]) std::swap(o[i], o[j]);
  double Vs[9];
  for (int k = 0; k < 3; ++k) {
    S[k] = sv[o[k]];
    for (int r = 0; r < 3; ++r) {
      Vs[r * 3 + k] = V[r * 3 + o[k]];
      U[r * 3 + k] = S[k] > 0 ? A[r * 3 + o[k]] / S[k] : 0.0;
    }
  }
  std::memcpy(V, Vs, sizeof(Vs));
  const double tiny = 1e-14 * (S[0] > 0 ? S[0] : 1.0);
  if (S[0] <= 0) {
    for (int i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return;
  }
  if (S[1] <= tiny) {  // rank 1: any unit vector orthogonal to u1
    const double u1[3] = {U[0], U[3], U[6]};
    int a = 0;
    if (std::fabs(u1[1]) < std::fabs(u1[a])) a = 1;
    if (std::fabs(u1[2]) < std::fabs(u1[a])) a = 2;
    double e[3] = {0, 0, 0};
    e[a] = 1.0;
    const double u2[3] = {u1[1] * e[2] - u1[2] * e[1], u1[2] * e[0] - u1[0] * e[2],
                          u1[0] * e[1] - u1[1] * e[0]};
    const double nn = std::sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    for (int r = 0; r < 3; ++r) U[r * 3 + 1] = u2[r] / nn;
  }
  if (S[2] <= tiny) {  // rank <= 2: u3 = u1 x u2
    U[2] = U[3] * U[7] - U[6] * U[4];
    U[5] = U[6] * U[1] - U[0] * U[7];
    U[8] = U[0] * U[4] - U[3] * U[1];
  }
}

inline double det3(const double M[9]) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
         M[2] * (M[3] * M[7] - M[4] * M[6]);
}

inline void rotation_from_correlation(const double H[9], double R[9]);

// weighted Kabsch without centring (the inputs are differences): R = V U^T of X diag(w) Y^T
inline void weighted_rotation(const std::vector<double>& X, const std::vector<double>& Y,
                              const std::vector<double>& w, double R[9]) {
  double H[9] = {0};
  const size_t n = w.size();
  for (size_t i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) H[a * 3 + b] += X[3 * i + a] * w[i] * Y[3 * i + b];
  rotation_from_correlation(H, R);
}

// R = V U^T of H = X diag(w) Y^T, reflection removed
inline void rotation_from_correlation(const double H[9], double R[9]) {
  double U[9], S[3], V[9];
  svd3(H, U, S, V);
  if (det3(U) * det3(V) < 0)
    for (int r = 0; r < 3; ++r) V[r * 3 + 2] = -V[r * 3 + 2];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b)
      R[a * 3 + b] = V[a * 3] * U[b * 3] + V[a * 3 + 1] * U[b * 3 + 1] + V[a * 3 + 2] * U[b * 3 + 2];
}

struct Params {
  double noise_bound = 0.5;          // IcpLoopComputation.cc:1022
  double cbar2 = 1.0;                // :1023
  int rotation_max_iterations = 100; // :1025
  double rotation_gnc_factor = 1.4;  // :1026
  double rotation_cost_threshold = 1e-6;  // :1030
};

// GNC-TLS: alternate the weighted rotation with closed-form weights while the surrogate's
// control parameter mu grows by the GNC factor
inline void gnc_rotation(const std::vector<double>& src, const std::vector<double>& dst,
                         const Params& p, double R[9]) {
  for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  const size_t n = src.size() / 3;
  if (n == 0) return;
  std::vector<double> w(n, 1.0), res(n);
  double mu = 1.0, prev_cost = INFINITY;
  double nb2 = p.noise_bound * p.noise_bound;
  if (nb2 < 1e-16) nb2 = 1e-2;
  for (int it = 0; it < p.rotation_max_iterations; ++it) {
    weighted_rotation(src, dst, w, R);
    double max_res = 0;
    for (size_t i = 0; i < n; ++i) {
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
      if (mu <= 0) break;
    }
    const double th1 = (mu + 1) / mu * nb2, th2 = mu / (mu + 1) * nb2;
    double cost = 0;
    for (size_t i = 0; i < n; ++i) {
      cost += w[i] * res[i];
      if (res[i] >= th1)
        w[i] = 0;
      else if (res[i] <= th2)
        w[i] = 1;
      else
        w[i] = std::sqrt(nb2 * mu * (mu + 1) / res[i]) - mu;
    }
    const double cost_diff = std::fabs(cost - prev_cost);
    mu *= p.rotation_gnc_factor;
    prev_cost = cost;
    if (cost_diff < p.rotation_cost_threshold) break;
  }
}

// truncated least squares of scalars with a common inlier range: sweep the 2n interval ends in
// ascending order, keep the consensus set's running sums, take the cheapest candidate
inline double tls_scalar(const std::vector<double>& X, double range) {
  struct End {
    double v;
    int id, seq;
  };
  const int n = (int)X.size();
  std::vector<End> h(2 * (size_t)n);
  for (int i = 0; i < n; ++i) {
    h[2 * i] = {X[i] - range, i + 1, 2 * i};
    h[2 * i + 1] = {X[i] + range, -i - 1, 2 * i + 1};
  }
  std::sort(h.begin(), h.end(),
            [](const End& a, const End& b) { return a.v < b.v || (a.v == b.v && a.seq < b.seq); });
  const double weight = 1.0 / (range * range);
  double ranges_sum = range * n, dot_xw = 0, dot_w = 0, sum_x = 0, sum_xx = 0;
  int card = 0;
  bool have = false;
  double best_cost = 0, best_x = 0;
  for (const End& e : h) {
    const int idx = std::abs(e.id) - 1, eps = e.id > 0 ? 1 : -1;
    card += eps;
    dot_w += eps * weight;
    dot_xw += eps * weight * X[idx];
    ranges_sum -= eps * range;
    sum_x += eps * X[idx];
    sum_xx += eps * X[idx] * X[idx];
    if (card <= 0) continue;
    const double x_hat = dot_xw / dot_w;
    const double cost = (card * x_hat * x_hat + sum_xx - 2 * sum_x * x_hat) + ranges_sum;
    if (!have || cost < best_cost) have = true, best_cost = cost, best_x = x_hat;
  }
  return best_x;
}

struct Solution {
  bool valid = false;
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double t[3] = {0, 0, 0};
  std::vector<int> clique;
};

// rotation from the clique's matches (cs/cd are [m][3], ascending match ids).  The library's
// default TIM graph for the rotation stage is the chain (Params::rotation_tim_graph = CHAIN, which
// the call site leaves alone): member i -> member i + 1, the last one back to the first; m
// differences, so the whole GNC-TLS loop is a few thousand operations per iteration.
inline void clique_rotation_host(const std::vector<double>& cs, const std::vector<double>& cd,
                                 const Params& p, double R[9]) {
  const int m = (int)(cs.size() / 3);
  std::vector<double> st((size_t)3 * m), dt((size_t)3 * m);
  for (int i = 0; i < m; ++i) {
    const int leaf = i + 1 < m ? i + 1 : 0;
    for (int c = 0; c < 3; ++c) {
      st[3 * (size_t)i + c] = cs[3 * leaf + c] - cs[3 * i + c];
      dt[3 * (size_t)i + c] = cd[3 * leaf + c] - cd[3 * i + c];
    }
  }
  gnc_rotation(st, dt, p, R);
}

// the matches of a clique, [m][3] each
inline void clique_points(const std::vector<double>& src, const std::vector<double>& dst,
                          const std::vector<int>& clique, std::vector<double>* cs, std::vector<double>* cd) {
  const int m = (int)clique.size();
  cs->resize((size_t)3 * m), cd->resize((size_t)3 * m);
  for (int i = 0; i < m; ++i)
    for (int c = 0; c < 3; ++c) {
      (*cs)[3 * i + c] = src[3 * clique[i] + c];
      (*cd)[3 * i + c] = dst[3 * clique[i] + c];
    }
}

// per-axis truncated-least-squares vote over the clique's matches under rotation R
inline void clique_translation(const std::vector<double>& cs, const std::vector<double>& cd, const double R[9],
                               const Params& p, double t[3]) {
  const int m = (int)(cs.size() / 3);
  std::vector<double> v(m);
  for (int a = 0; a < 3; ++a) {
    for (int i = 0; i < m; ++i) {
      const double* x = &cs[3 * i];
      v[i] = cd[3 * i + a] - (R[a * 3] * x[0] + R[a * 3 + 1] * x[1] + R[a * 3 + 2] * x[2]);
    }
    t[a] = tls_scalar(v, p.cbar2 * p.noise_bound);
  }
}

// rotation and translation from the matches of the clique (src/dst are [n][3]); `rotation` is
// clique_rotation_host or a drop-in that does the same sums elsewhere
template <class RotationFn>
inline Solution solve_on_clique(const std::vector<double>& src, const std::vector<double>& dst,
                                std::vector<int> clique, const Params& p, RotationFn rotation) {
  Solution s;
  s.clique = std::move(clique);
  const int m = (int)s.clique.size();
  if (m <= 1) return s;
  std::vector<double> cs, cd;
  clique_points(src, dst, s.clique, &cs, &cd);
  rotation(cs, cd, p, s.R);
  clique_translation(cs, cd, s.R, p, s.t);
  s.valid = true;
  return s;
}

inline Solution solve_on_clique(const std::vector<double>& src, const std::vector<double>& dst,
                                std::vector<int> clique, const Params& p) {
  return solve_on_clique(src, dst, std::move(clique), p, clique_rotation_host);
}

// sum over matches of min(|R s + t - d|^2 / threshold, 1)
inline double error_metric(const std::vector<double>& src, const std::vector<double>& dst,
                           const double R[9], const double t[3], double threshold) {
  double err = 0;
  const size_t n = src.size() / 3;
  for (size_t i = 0; i < n; ++i) {
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

}  // namespace teaser
}  // namespace lgicp
