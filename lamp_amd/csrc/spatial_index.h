// spatial_index.h — host side of the spatial index: the Morton visiting order of a cloud.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

namespace lgicp {
namespace si {

inline uint64_t spread21(uint64_t v) {  // ..cba -> ..00c00b00a
  v &= 0x1FFFFFull;
  v = (v | (v << 32)) & 0x1F00000000FFFFull;
  v = (v | (v << 16)) & 0x1F0000FF0000FFull;
  v = (v | (v << 8)) & 0x100F00F00F00F00Full;
  v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

// 63-bit Morton code on a 1/64 m lattice covering +-16 km; non-finite points sort last
inline uint64_t morton(float x, float y, float z) {
  if (!std::isfinite(x) || !std::isfinite(y) || !std::isfinite(z)) return ~0ull;
  auto q = [](float v) {
    const double s = ((double)v + 16384.0) * 64.0;
    return (uint64_t)std::min(2097151.0, std::max(0.0, std::floor(s)));
  };
  return spread21(q(x)) | (spread21(q(y)) << 1) | (spread21(q(z)) << 2);
}

// permutation that visits the points in Morton order (ties by index); xyz has `stride` floats
// per point
inline std::vector<int> morton_order(const float* xyz, size_t n, size_t stride) {
  std::vector<std::pair<uint64_t, int>> k(n);
  for (size_t i = 0; i < n; ++i)
    k[i] = {morton(xyz[i * stride], xyz[i * stride + 1], xyz[i * stride + 2]), (int)i};
  std::sort(k.begin(), k.end());
  std::vector<int> order(n);
  for (size_t i = 0; i < n; ++i) order[i] = k[i].second;
  return order;
}

// Visiting order from a balanced k-d partition: the index range is halved at a tile boundary
// along the longest axis of its points' box (median split) until a range holds one tile.
// Consecutive runs of `leaf` points are then spatially compact with nearly disjoint boxes, and so
// are the runs of 2 * leaf, 4 * leaf ... points above them.  Non-finite points go last.
// Deterministic: ties along the split axis are ordered by index.
inline std::vector<int> kd_order(const float* xyz, size_t n, size_t stride, int leaf) {
  std::vector<int> order;
  order.reserve(n);
  std::vector<int> tail;
  for (size_t i = 0; i < n; ++i) {
    const float* p = xyz + i * stride;
    if (std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]))
      order.push_back((int)i);
    else
      tail.push_back((int)i);
  }
  struct Range {
    size_t lo, hi;
  };
  std::vector<Range> stack;
  if (!order.empty()) stack.push_back({0, order.size()});
  while (!stack.empty()) {
    const Range r = stack.back();
    stack.pop_back();
    const size_t m = r.hi - r.lo;
    if (m <= (size_t)leaf) continue;
    float lo3[3] = {INFINITY, INFINITY, INFINITY}, hi3[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = r.lo; i < r.hi; ++i) {
      const float* p = xyz + (size_t)order[i] * stride;
      for (int a = 0; a < 3; ++a) lo3[a] = std::min(lo3[a], p[a]), hi3[a] = std::max(hi3[a], p[a]);
    }
    int ax = 0;
    for (int a = 1; a < 3; ++a)
      if (hi3[a] - lo3[a] > hi3[ax] - lo3[ax]) ax = a;
    const size_t tiles = (m + leaf - 1) / leaf;
    const size_t k = (tiles / 2) * (size_t)leaf;  // left half: whole tiles
    std::nth_element(order.begin() + r.lo, order.begin() + r.lo + k, order.begin() + r.hi,
                     [&](int a, int b) {
                       const float va = xyz[(size_t)a * stride + ax], vb = xyz[(size_t)b * stride + ax];
                       return va < vb || (va == vb && a < b);
                     });
    stack.push_back({r.lo, r.lo + k});
    stack.push_back({r.lo + k, r.hi});
  }
  order.insert(order.end(), tail.begin(), tail.end());
  return order;
}

}  // namespace si
}  // namespace lgicp
