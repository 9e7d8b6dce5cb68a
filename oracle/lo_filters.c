/* lo_filters.c — CPU restatement of the keyed-scan down-sampling filters: pcl::VoxelGrid and
 * pcl::RandomSample as LampPcldFilter drives them (lamp_utils/src/LampPcldFilter.cc:17-72;
 * the evaluation driver repeats both, loop_closure/src/eval_loop_computation_test.cc:133-251).
 * TEST INFRASTRUCTURE ONLY.
 *
 * PCL is not part of the reference tree and no reference test pins a filter output: parity
 * unpinned.  pcl::VoxelGrid::applyFilter (PCL 1.8) is restated from its published source:
 * bounding box of the finite points, inverse leaf size in float, the 64-bit check that the
 * grid has at most INT_MAX cells (else the input is handed back), cell coordinates
 * floor(p * inv) - min_b, linear index with multipliers (1, nx, nx * ny), points sorted by
 * index, one centroid per run: float sums divided by the count, normals summed and normalised
 * (CentroidPoint).  std::sort leaves the order inside a run unspecified; ascending input order
 * is used here.  pcl::RandomSample draws with rand() seeded from the wall clock; what is
 * restated is its contract (n_keep points, input order) with a draw that is a pure function
 * of a seed: the n_keep smallest (hash(seed, index), index) pairs. */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "lamp_oracle.h"

typedef struct {
  uint32_t key;
  int32_t id;
} KeyId;

static int cmp_keyid(const void* a, const void* b) {
  const KeyId* x = (const KeyId*)a;
  const KeyId* y = (const KeyId*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}

static int finite3(const float* p) { return isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]); }

size_t lo_voxel_grid(const float* xyz, const float* normals, size_t n, double leaf_size,
                     float* out_xyz, float* out_normals) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  size_t n_fin = 0;
  for (size_t i = 0; i < n; ++i) {
    if (!finite3(xyz + 3 * i)) continue;
    ++n_fin;
    for (int a = 0; a < 3; ++a) {
      if (xyz[3 * i + a] < lo[a]) lo[a] = xyz[3 * i + a];
      if (xyz[3 * i + a] > hi[a]) hi[a] = xyz[3 * i + a];
    }
  }
  if (n_fin == 0) return 0;
  const float inv = 1.0f / (float)leaf_size;
  int min_b[3], div_b[3];
  int64_t d[3];
  for (int a = 0; a < 3; ++a) {
    d[a] = (int64_t)((hi[a] - lo[a]) * inv) + 1;
    min_b[a] = (int)floorf(lo[a] * inv);
    div_b[a] = (int)floorf(hi[a] * inv) - min_b[a] + 1;
  }
  /* overflow-free form of dx * dy * dz > INT_MAX */
  if (d[0] > INT_MAX || d[1] > INT_MAX || d[2] > INT_MAX || d[0] * d[1] > INT_MAX ||
      d[0] * d[1] * d[2] > INT_MAX) {
    memcpy(out_xyz, xyz, sizeof(float) * 3 * n);
    if (normals && out_normals) memcpy(out_normals, normals, sizeof(float) * 3 * n);
    return n;
  }
  const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
  KeyId* v = (KeyId*)malloc(sizeof(KeyId) * n_fin);
  size_t m = 0;
  for (size_t i = 0; i < n; ++i) {
    const float* p = xyz + 3 * i;
    if (!finite3(p)) continue;
    const int i0 = (int)(floorf(p[0] * inv) - (float)min_b[0]);
    const int i1 = (int)(floorf(p[1] * inv) - (float)min_b[1]);
    const int i2 = (int)(floorf(p[2] * inv) - (float)min_b[2]);
    v[m].key = (uint32_t)(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]);
    v[m].id = (int32_t)i;
    ++m;
  }
  qsort(v, m, sizeof(KeyId), cmp_keyid);
  size_t n_out = 0;
  for (size_t i = 0; i < m;) {
    size_t j = i;
    float s[3] = {0.f, 0.f, 0.f}, sn[3] = {0.f, 0.f, 0.f};
    for (; j < m && v[j].key == v[i].key; ++j) {
      const size_t p = 3 * (size_t)v[j].id;
      for (int a = 0; a < 3; ++a) {
        s[a] += xyz[p + a];
        if (normals) sn[a] += normals[p + a];
      }
    }
    const float c = (float)(j - i);
    for (int a = 0; a < 3; ++a) out_xyz[3 * n_out + a] = s[a] / c;
    if (normals && out_normals) {
      const float z = (sn[0] * sn[0] + sn[1] * sn[1]) + sn[2] * sn[2];
      if (z > 0.f) {
        const float r = sqrtf(z);
        for (int a = 0; a < 3; ++a) sn[a] /= r;
      }
      for (int a = 0; a < 3; ++a) out_normals[3 * n_out + a] = sn[a];
    }
    ++n_out;
    i = j;
  }
  free(v);
  return n_out;
}

static uint32_t sample_hash(uint32_t seed, uint32_t i) {
  uint32_t h = ((i + 0x9E3779B9u) * 0x85EBCA6Bu) ^ (seed * 0xC2B2AE35u + 0x27D4EB2Fu);
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

static int cmp_i32(const void* a, const void* b) {
  const int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

size_t lo_random_sample(size_t n, size_t n_keep, uint32_t seed, int32_t* out_indices) {
  if (n_keep >= n) {
    for (size_t i = 0; i < n; ++i) out_indices[i] = (int32_t)i;
    return n;
  }
  if (n_keep == 0) return 0;
  KeyId* v = (KeyId*)malloc(sizeof(KeyId) * n);
  for (size_t i = 0; i < n; ++i) {
    v[i].key = sample_hash(seed, (uint32_t)i);
    v[i].id = (int32_t)i;
  }
  qsort(v, n, sizeof(KeyId), cmp_keyid);
  for (size_t i = 0; i < n_keep; ++i) out_indices[i] = v[i].id;
  qsort(out_indices, n_keep, sizeof(int32_t), cmp_i32);
  free(v);
  return n_keep;
}
