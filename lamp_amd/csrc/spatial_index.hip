// spatial_index.hip — builds the tile/box index the pruned 1-NN kernel walks.
//
// The visiting order (k-d partition) of a keyframe's points is computed once on the host when the
// keyframe is registered (spatial_index.h); this kernel applies it: it gathers the points of one
// segment into tiles of SI_TILE, tags each with its original index inside the (possibly
// accumulated) cloud, and reduces each tile's axis-aligned box.  An accumulated cloud
// (IcpLoopComputation::AccumulateScans, loop_closure/src/IcpLoopComputation.cc:910-947) is
// indexed as the concatenation of its keyframes' segments: tiles only have to be compact, not
// globally sorted.
#include "lgicp_device.h"

namespace lgicp {

__global__ __launch_bounds__(64) void index_fill_kernel(const IndexSegDev* __restrict__ segs) {
  const IndexSegDev& G = segs[blockIdx.y];
  const int lane = threadIdx.x;
  const int n_tiles = (G.n + SI_TILE - 1) / SI_TILE;
  const float inf = __builtin_inff();
  constexpr int TPW = 64 / SI_TILE;  // tiles per wave
  const int sub = lane % SI_TILE;
  for (int t0 = blockIdx.x * TPW; t0 < n_tiles; t0 += gridDim.x * TPW) {
    const int t = t0 + lane / SI_TILE;
    const int s = t * SI_TILE + sub;
    float4 p = make_float4(inf, inf, inf, __int_as_float(-1));
    if (t < n_tiles && s < G.n) {
      const int orig = G.point_off + G.order[s];
      const float4 q = G.xyz[orig];
      p = make_float4(q.x, q.y, q.z, __int_as_float(orig));
      G.qorder[G.point_off + s] = orig;
    }
    if (t < n_tiles) {  // one tile = x[SI_TILE] y[SI_TILE] z[SI_TILE] index[SI_TILE]
      float* dst = reinterpret_cast<float*>(G.sxyz) + (size_t)(G.tile_off + t) * (4 * SI_TILE) + sub;
      dst[0] = p.x, dst[SI_TILE] = p.y, dst[2 * SI_TILE] = p.z, dst[3 * SI_TILE] = p.w;
    }
    // box over the finite points of the tile (fminf/fmaxf drop NaN operands)
    const bool fin = __builtin_isfinite(p.x) && __builtin_isfinite(p.y) && __builtin_isfinite(p.z);
    const float nan = __builtin_nanf("");
    float x0 = fin ? p.x : nan, y0 = fin ? p.y : nan, z0 = fin ? p.z : nan;
    float x1 = x0, y1 = y0, z1 = z0;
#pragma unroll
    for (int off = SI_TILE / 2; off > 0; off >>= 1) {
      x0 = __builtin_fminf(x0, __shfl_xor(x0, off));
      y0 = __builtin_fminf(y0, __shfl_xor(y0, off));
      z0 = __builtin_fminf(z0, __shfl_xor(z0, off));
      x1 = __builtin_fmaxf(x1, __shfl_xor(x1, off));
      y1 = __builtin_fmaxf(y1, __shfl_xor(y1, off));
      z1 = __builtin_fmaxf(z1, __shfl_xor(z1, off));
    }
    if (t < n_tiles && sub == 0) {
      // a tile without a finite point gets an empty box at infinity: its gap to any query is +inf
      const bool any = !__builtin_isnan(x0);
      G.tmin[G.tile_off + t] = any ? make_float4(x0, y0, z0, 0.f) : make_float4(inf, inf, inf, 0.f);
      G.tmax[G.tile_off + t] = any ? make_float4(x1, y1, z1, 0.f) : make_float4(inf, inf, inf, 0.f);
    }
  }
  if (G.last && blockIdx.x == 0)
    for (int i = G.point_off + G.n + lane; i < G.n_q; i += 64) G.qorder[i] = -1;
}

// Boxes of the blocks of 64 consecutive tiles of a cloud, stored behind the cloud's tile boxes
// (entries n_tiles .. n_tiles + ceil(n_tiles / 64) - 1 of tmin / tmax).  In k-d order 64
// consecutive tiles are a compact piece of the cloud, so a search looks at these boxes first and
// at the tile boxes of the few blocks that come close.  Runs after index_fill_kernel on the last
// segment of every cloud (all tile boxes of the cloud are in place by then).
__global__ __launch_bounds__(64) void index_blocks_kernel(const IndexSegDev* __restrict__ segs) {
  const IndexSegDev& G = segs[blockIdx.y];
  if (!G.last) return;
  const int lane = threadIdx.x;
  const int n_tiles = G.tile_off + (G.n + SI_TILE - 1) / SI_TILE;
  const int n_blocks = (n_tiles + 63) / 64;
  const float inf = __builtin_inff();
  for (int b = blockIdx.x; b < n_blocks; b += gridDim.x) {
    const int t = b * 64 + lane;
    float x0 = inf, y0 = inf, z0 = inf, x1 = -inf, y1 = -inf, z1 = -inf;
    if (t < n_tiles) {
      const float4 a = G.tmin[t], c = G.tmax[t];
      if (a.x < inf) x0 = a.x, y0 = a.y, z0 = a.z, x1 = c.x, y1 = c.y, z1 = c.z;  // not an empty tile
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      x0 = __builtin_fminf(x0, __shfl_xor(x0, off));
      y0 = __builtin_fminf(y0, __shfl_xor(y0, off));
      z0 = __builtin_fminf(z0, __shfl_xor(z0, off));
      x1 = __builtin_fmaxf(x1, __shfl_xor(x1, off));
      y1 = __builtin_fmaxf(y1, __shfl_xor(y1, off));
      z1 = __builtin_fmaxf(z1, __shfl_xor(z1, off));
    }
    if (lane == 0) {
      const bool any = x0 < inf;  // otherwise an empty box at infinity, like an empty tile's
      G.tmin[n_tiles + b] = any ? make_float4(x0, y0, z0, 0.f) : make_float4(inf, inf, inf, 0.f);
      G.tmax[n_tiles + b] = any ? make_float4(x1, y1, z1, 0.f) : make_float4(inf, inf, inf, 0.f);
    }
  }
}

hipError_t launch_index_fill(const IndexSegDev* segs, int n_segs, int max_n, hipStream_t stream) {
  if (n_segs <= 0) return hipSuccess;
  int bx = (max_n + 63) / 64;
  bx = bx < 1 ? 1 : (bx > 256 ? 256 : bx);
  hipLaunchKernelGGL(index_fill_kernel, dim3(bx, n_segs), dim3(64), 0, stream, segs);
  hipLaunchKernelGGL(index_blocks_kernel, dim3(16, n_segs), dim3(64), 0, stream, segs);
  return hipGetLastError();
}

}  // namespace lgicp
