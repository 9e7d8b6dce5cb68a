/*
 * lo_kdtree.c — exact nearest-neighbour search for the CPU oracle (TEST INFRASTRUCTURE
 * ONLY, see lamp_oracle.h).
 *
 * The reference searches with pcl::search::KdTree -> FLANN KDTreeSingleIndex<L2_Simple<float>>
 * (leaf 15, exact; call sites lamp_utils/include/lamp_utils/gicp.h:390-397,
 * gicp.hpp:108-109, loop_closure/src/IcpLoopComputation.cc:555-564).  FLANN is not under
 * /root/reference; what is restated here is its published distance arithmetic
 * (result += diff*diff over x, y, z in that order, float, no FMA) and exactness.  FLANN's
 * winner among exactly equidistant targets depends on tree traversal order; this oracle
 * defines the tie rule as "lowest target index wins", for the brute-force and the kd-tree
 * search alike, so both return identical index arrays.
 */
#define _GNU_SOURCE
#include "lamp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline float dist2(const float* a, const float* b) {
  /* FLANN L2_Simple: result = 0; result += diff*diff for each dimension */
  float d = a[0] - b[0];
  float r = d * d;
  d = a[1] - b[1];
  r = r + d * d;
  d = a[2] - b[2];
  r = r + d * d;
  return r;
}

void lo_nn_brute(const float* q, int nq, const float* t, int nt, int32_t* idx, float* d2) {
  for (int i = 0; i < nq; ++i) {
    float best = INFINITY;
    int32_t bi = -1;
    for (int j = 0; j < nt; ++j) {
      const float d = dist2(q + 3 * i, t + 3 * j);
      if (d < best) {
        best = d;
        bi = j;
      }
    }
    idx[i] = bi;
    d2[i] = best;
  }
}

/* insertion into an ascending (d, idx) list of length k */
static inline void knn_insert(float* bd, int32_t* bi, int k, float d, int32_t j) {
  if (!(d < bd[k - 1] || (d == bd[k - 1] && j < bi[k - 1]))) return;
  int p = k - 1;
  while (p > 0 && (d < bd[p - 1] || (d == bd[p - 1] && j < bi[p - 1]))) {
    bd[p] = bd[p - 1];
    bi[p] = bi[p - 1];
    --p;
  }
  bd[p] = d;
  bi[p] = j;
}

void lo_knn_brute(const float* q, int nq, const float* t, int nt, int k, int32_t* idx, float* d2) {
  for (int i = 0; i < nq; ++i) {
    float* bd = d2 + (size_t)i * k;
    int32_t* bi = idx + (size_t)i * k;
    for (int p = 0; p < k; ++p) {
      bd[p] = INFINITY;
      bi[p] = INT32_MAX;
    }
    for (int j = 0; j < nt; ++j) knn_insert(bd, bi, k, dist2(q + 3 * i, t + 3 * j), j);
    for (int p = 0; p < k; ++p)
      if (bi[p] == INT32_MAX) bi[p] = -1;
  }
}

/* ------------------------------------------------------------------ kd-tree */
#define LO_LEAF 15

typedef struct {
  int32_t left, right; /* children (node ids) or -1 for a leaf          */
  int32_t begin, end;  /* leaf: range in perm                            */
  int32_t dim;
  float split_lo, split_hi; /* max of left child / min of right child along dim */
} lo_node;

struct lo_kdtree {
  const float* xyz; /* owned copy */
  float* xyz_own;
  int32_t* perm;
  lo_node* nodes;
  int n, n_nodes, cap_nodes;
};

static int32_t new_node(lo_kdtree* t) {
  if (t->n_nodes == t->cap_nodes) {
    t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 64;
    t->nodes = (lo_node*)realloc(t->nodes, sizeof(lo_node) * (size_t)t->cap_nodes);
  }
  return t->n_nodes++;
}

typedef struct {
  const float* xyz;
  int dim;
} lo_sort_ctx;
static int cmp_perm(const void* a, const void* b, void* arg) {
  const lo_sort_ctx* c = (const lo_sort_ctx*)arg;
  const int32_t ia = *(const int32_t*)a, ib = *(const int32_t*)b;
  const float va = c->xyz[3 * ia + c->dim], vb = c->xyz[3 * ib + c->dim];
  if (va < vb) return -1;
  if (va > vb) return 1;
  return (ia > ib) - (ia < ib);
}

static int32_t build_rec(lo_kdtree* t, int begin, int end) {
  const int32_t id = new_node(t);
  lo_node nd;
  nd.left = nd.right = -1;
  nd.begin = begin;
  nd.end = end;
  nd.dim = 0;
  nd.split_lo = nd.split_hi = 0.f;
  if (end - begin <= LO_LEAF) {
    t->nodes[id] = nd;
    return id;
  }
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = begin; i < end; ++i)
    for (int d = 0; d < 3; ++d) {
      const float v = t->xyz[3 * t->perm[i] + d];
      if (v < lo[d]) lo[d] = v;
      if (v > hi[d]) hi[d] = v;
    }
  int dim = 0;
  if (hi[1] - lo[1] > hi[dim] - lo[dim]) dim = 1;
  if (hi[2] - lo[2] > hi[dim] - lo[dim]) dim = 2;
  lo_sort_ctx ctx = {t->xyz, dim};
  qsort_r(t->perm + begin, (size_t)(end - begin), sizeof(int32_t), cmp_perm, &ctx);
  const int mid = (begin + end) / 2;
  nd.dim = dim;
  nd.split_lo = t->xyz[3 * t->perm[mid - 1] + dim];
  nd.split_hi = t->xyz[3 * t->perm[mid] + dim];
  t->nodes[id] = nd;
  const int32_t l = build_rec(t, begin, mid);
  const int32_t r = build_rec(t, mid, end);
  t->nodes[id].left = l;
  t->nodes[id].right = r;
  return id;
}

lo_kdtree* lo_kdtree_build(const float* xyz, int n) {
  lo_kdtree* t = (lo_kdtree*)calloc(1, sizeof(lo_kdtree));
  t->n = n;
  t->xyz_own = (float*)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
  memcpy(t->xyz_own, xyz, sizeof(float) * 3 * (size_t)n);
  t->xyz = t->xyz_own;
  t->perm = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) t->perm[i] = i;
  if (n > 0) build_rec(t, 0, n);
  return t;
}

void lo_kdtree_free(lo_kdtree* t) {
  if (!t) return;
  free(t->xyz_own);
  free(t->perm);
  free(t->nodes);
  free(t);
}

typedef struct {
  float* bd;
  int32_t* bi;
  int k;
} lo_heap;

/* Branch pruning uses per-dimension plane distances, which never exceed the true distance
 * in exact arithmetic; the test is `>` (not `>=`) so that equidistant candidates in the far
 * branch are still visited and the lowest-index rule holds. */
static void search_rec(const lo_kdtree* t, int32_t id, const float* q, lo_heap* h) {
  const lo_node* nd = &t->nodes[id];
  if (nd->left < 0) {
    for (int i = nd->begin; i < nd->end; ++i) {
      const int32_t j = t->perm[i];
      knn_insert(h->bd, h->bi, h->k, dist2(q, t->xyz + 3 * j), j);
    }
    return;
  }
  const float v = q[nd->dim];
  const float dl = v - nd->split_lo; /* >0: query is right of the left child's max  */
  const float dh = nd->split_hi - v; /* >0: query is left of the right child's min   */
  int32_t first, second;
  float plane;
  if (dl <= dh || !(v == v)) {
    first = nd->left;
    second = nd->right;
    plane = dh > 0.f ? dh * dh : 0.f;
  } else {
    first = nd->right;
    second = nd->left;
    plane = dl > 0.f ? dl * dl : 0.f;
  }
  search_rec(t, first, q, h);
  if (!(plane > h->bd[h->k - 1])) search_rec(t, second, q, h);
}

void lo_kdtree_knn(const lo_kdtree* t, const float* q, int nq, int k, int32_t* idx, float* d2,
                   int threads) {
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1) if (threads > 1)
#endif
  for (int i = 0; i < nq; ++i) {
    lo_heap h;
    h.bd = d2 + (size_t)i * k;
    h.bi = idx + (size_t)i * k;
    h.k = k;
    for (int p = 0; p < k; ++p) {
      h.bd[p] = INFINITY;
      h.bi[p] = INT32_MAX;
    }
    if (t->n > 0) search_rec(t, 0, q + 3 * i, &h);
    for (int p = 0; p < k; ++p)
      if (h.bi[p] == INT32_MAX) h.bi[p] = -1;
  }
}

void lo_kdtree_nn(const lo_kdtree* t, const float* q, int nq, int32_t* idx, float* d2, int threads) {
  lo_kdtree_knn(t, q, nq, 1, idx, d2, threads);
}
