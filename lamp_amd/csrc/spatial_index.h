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

}  // namespace si
}  // namespace lgicp
