// cloud_filters.hip — keyed-scan down-sampling on the device: voxel grid and random sample.
//
// Replaces the two PCL filters behind LampPcldFilter (lamp_utils/src/LampPcldFilter.cc:17-72)
// and the evaluation driver's filters (loop_closure/src/eval_loop_computation_test.cc:133-251):
// pcl::VoxelGrid with a cubic leaf (all fields averaged) and pcl::RandomSample.
//
// Voxel grid, as PCL computes it: bounding box of the finite points, integer cell coordinates
// floor(p * (1 / leaf)) - floor(min * (1 / leaf)) in float, linear cell index x + nx (y + nz z),
// points sorted by cell index, one output point per occupied cell in ascending cell order:
// the mean of its points (float sums, divided by the count) and the normalised sum of their
// normals.  PCL orders the points of one cell however std::sort leaves them; here they are
// summed in ascending input order (the radix sort is stable), so outputs can differ from PCL's
// in the last float bits.  All of it is HBM-bound integer/byte work: one pass for the bounds,
// one for the keys, a radix sort of (cell, point) pairs, a scan over the run heads and one
// gather pass for the means.
//
// Random sample: PCL draws with rand() seeded by the wall clock, so only the contract is
// restated: exactly n_keep points, input order kept.  Every point gets a 32-bit hash of
// (seed, index); the n_keep smallest (hash, index) pairs are the sample.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "lgicp_internal.h"

namespace lgicp {

constexpr unsigned VOXEL_NONE = 0xFFFFFFFFu;  // key of a non-finite point: sorts last, owns no cell
constexpr int FILTER_THREADS = 256;

// per-block bounding box of the finite points; part[b] = {min x y z, max x y z}
__global__ __launch_bounds__(FILTER_THREADS) void bounds_kernel(const float* __restrict__ xyz, int n,
                                                                float* __restrict__ part) {
  __shared__ float sh[FILTER_THREADS / 64][6];
  float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
  float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (int i = blockIdx.x * FILTER_THREADS + threadIdx.x; i < n; i += gridDim.x * FILTER_THREADS) {
    const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
    if (!(__builtin_isfinite(x) && __builtin_isfinite(y) && __builtin_isfinite(z))) continue;
    lo[0] = fminf(lo[0], x), lo[1] = fminf(lo[1], y), lo[2] = fminf(lo[2], z);
    hi[0] = fmaxf(hi[0], x), hi[1] = fmaxf(hi[1], y), hi[2] = fmaxf(hi[2], z);
  }
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
    }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
#pragma unroll
    for (int a = 0; a < 3; ++a) sh[wave][a] = lo[a], sh[wave][3 + a] = hi[a];
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = sh[0][threadIdx.x];
    for (int w = 1; w < FILTER_THREADS / 64; ++w)
      v = threadIdx.x < 3 ? fminf(v, sh[w][threadIdx.x]) : fmaxf(v, sh[w][threadIdx.x]);
    part[blockIdx.x * 6 + threadIdx.x] = v;
  }
}

struct VoxelFrame {
  float inv;      // 1 / leaf in float
  int min_b[3];   // floor(min * inv)
  int mul[3];     // 1, nx, nx * ny
};

__global__ __launch_bounds__(FILTER_THREADS) void voxel_key_kernel(const float* __restrict__ xyz,
                                                                   int n, VoxelFrame f,
                                                                   unsigned* __restrict__ keys,
                                                                   int* __restrict__ ids) {
  const int i = blockIdx.x * FILTER_THREADS + threadIdx.x;
  if (i >= n) return;
  const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
  unsigned key = VOXEL_NONE;
  if (__builtin_isfinite(x) && __builtin_isfinite(y) && __builtin_isfinite(z)) {
    const int i0 = (int)(floorf(x * f.inv) - (float)f.min_b[0]);
    const int i1 = (int)(floorf(y * f.inv) - (float)f.min_b[1]);
    const int i2 = (int)(floorf(z * f.inv) - (float)f.min_b[2]);
    key = (unsigned)(i0 * f.mul[0] + i1 * f.mul[1] + i2 * f.mul[2]);
  }
  keys[i] = key;
  ids[i] = i;
}

// head[i] = 1 where a run of equal cell keys starts
__global__ __launch_bounds__(FILTER_THREADS) void run_head_kernel(const unsigned* __restrict__ keys,
                                                                  int n, int* __restrict__ head) {
  const int i = blockIdx.x * FILTER_THREADS + threadIdx.x;
  if (i >= n) return;
  const unsigned k = keys[i];
  head[i] = (k != VOXEL_NONE && (i == 0 || keys[i - 1] != k)) ? 1 : 0;
}

// one lane per run head: walks its run in sorted (= ascending input) order
__global__ __launch_bounds__(FILTER_THREADS) void voxel_mean_kernel(
    const float* __restrict__ xyz, const float* __restrict__ nrm, const unsigned* __restrict__ keys,
    const int* __restrict__ ids, const int* __restrict__ head, const int* __restrict__ slot, int n,
    float* __restrict__ out_xyz, float* __restrict__ out_nrm) {
  const int i = blockIdx.x * FILTER_THREADS + threadIdx.x;
  if (i >= n || !head[i]) return;
  const unsigned k = keys[i];
  float sx = 0.f, sy = 0.f, sz = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
  int cnt = 0;
  for (int j = i; j < n && keys[j] == k; ++j) {
    const size_t p = 3 * (size_t)ids[j];
    sx += xyz[p], sy += xyz[p + 1], sz += xyz[p + 2];
    if (nrm) nx += nrm[p], ny += nrm[p + 1], nz += nrm[p + 2];
    ++cnt;
  }
  const size_t o = 3 * (size_t)slot[i];
  const float c = (float)cnt;
  out_xyz[o] = sx / c, out_xyz[o + 1] = sy / c, out_xyz[o + 2] = sz / c;
  if (out_nrm) {
    const float z = (nx * nx + ny * ny) + nz * nz;
    if (z > 0.f) {  // a zero sum stays zero
      const float s = sqrtf(z);
      nx /= s, ny /= s, nz /= s;
    }
    out_nrm[o] = nx, out_nrm[o + 1] = ny, out_nrm[o + 2] = nz;
  }
}

__device__ __host__ inline unsigned sample_hash(unsigned seed, unsigned i) {
  unsigned h = (i + 0x9E3779B9u) * 0x85EBCA6Bu ^ (seed * 0xC2B2AE35u + 0x27D4EB2Fu);
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

__global__ __launch_bounds__(FILTER_THREADS) void sample_key_kernel(int n, unsigned seed,
                                                                    unsigned* __restrict__ keys,
                                                                    int* __restrict__ ids) {
  const int i = blockIdx.x * FILTER_THREADS + threadIdx.x;
  if (i >= n) return;
  keys[i] = sample_hash(seed, (unsigned)i);
  ids[i] = i;
}

__global__ __launch_bounds__(FILTER_THREADS) void gather3_kernel(const float* __restrict__ in,
                                                                 const unsigned* __restrict__ ids,
                                                                 int n, float* __restrict__ out) {
  const int i = blockIdx.x * FILTER_THREADS + threadIdx.x;
  if (i >= n) return;
  const size_t p = 3 * (size_t)ids[i];
  out[3 * (size_t)i] = in[p], out[3 * (size_t)i + 1] = in[p + 1], out[3 * (size_t)i + 2] = in[p + 2];
}

}  // namespace lgicp

using namespace lgicp_detail;
using namespace lgicp;

namespace {

struct Dev {  // frees everything it handed out
  std::vector<void*> ptrs;
  ~Dev() {
    for (void* p : ptrs) (void)hipFree(p);
  }
  template <typename T>
  T* alloc(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
    ptrs.push_back(p);
    return (T*)p;
  }
};

inline dim3 blocks_for(size_t n) { return dim3((unsigned)((n + FILTER_THREADS - 1) / FILTER_THREADS)); }

int copy_through(const float* xyz, const float* normals, size_t n, float* out_xyz, float* out_normals,
                 size_t* n_out) {
  std::memcpy(out_xyz, xyz, sizeof(float) * 3 * n);
  if (normals && out_normals) std::memcpy(out_normals, normals, sizeof(float) * 3 * n);
  *n_out = n;
  return LGICP_OK;
}

}  // namespace

extern "C" {

int lgicp_voxel_grid(lgicp_ctx* ctx, const float* xyz, const float* normals, size_t n,
                     double leaf_size, float* out_xyz, float* out_normals, size_t* n_out) {
  if (!ctx || !n_out || (n && (!xyz || !out_xyz)) || (out_normals && !normals))
    return fail(ctx, LGICP_E_INVALID, "null argument");
  if (!(leaf_size > 0)) return fail(ctx, LGICP_E_INVALID, "leaf size must be positive");
  if (n > (size_t)1 << 30) return fail(ctx, LGICP_E_INVALID, "cloud too large");
  *n_out = 0;
  if (n == 0) return LGICP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t st = ctx->stream;
  Dev d;
  const int ni = (int)n;
  const int n_bound_blocks = (int)std::min<size_t>(1024, (n + FILTER_THREADS - 1) / FILTER_THREADS);
  float* d_xyz = d.alloc<float>(3 * n);
  float* d_nrm = normals ? d.alloc<float>(3 * n) : nullptr;
  float* d_part = d.alloc<float>((size_t)6 * n_bound_blocks);
  if (!d_xyz || (normals && !d_nrm) || !d_part) return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  HIP_TRY(ctx, hipMemcpyAsync(d_xyz, xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, st));
  if (normals)
    HIP_TRY(ctx, hipMemcpyAsync(d_nrm, normals, sizeof(float) * 3 * n, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(bounds_kernel, dim3(n_bound_blocks), dim3(FILTER_THREADS), 0, st, d_xyz, ni, d_part);
  HIP_TRY(ctx, hipGetLastError());
  std::vector<float> part((size_t)6 * n_bound_blocks);
  HIP_TRY(ctx, hipMemcpyAsync(part.data(), d_part, sizeof(float) * part.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int b = 0; b < n_bound_blocks; ++b)
    for (int a = 0; a < 3; ++a) {
      lo[a] = std::fmin(lo[a], part[6 * b + a]);
      hi[a] = std::fmax(hi[a], part[6 * b + 3 + a]);
    }
  if (!(lo[0] <= hi[0])) return LGICP_OK;  // no finite point
  // pcl::VoxelGrid::applyFilter: float inverse leaf, 64-bit cell-count check, then int frames
  VoxelFrame f;
  f.inv = 1.0f / (float)leaf_size;
  long long cells = 1;
  int div_b[3];
  for (int a = 0; a < 3; ++a) {
    const long long da = (long long)((hi[a] - lo[a]) * f.inv) + 1;
    cells = (cells > (1ll << 40) || da > (1ll << 40)) ? (1ll << 62) : cells * da;
    f.min_b[a] = (int)std::floor(lo[a] * f.inv);
    div_b[a] = (int)std::floor(hi[a] * f.inv) - f.min_b[a] + 1;
  }
  if (cells > (long long)std::numeric_limits<int32_t>::max())
    // "Leaf size is too small for the input dataset. Integer indices would overflow.": PCL
    // warns and hands the input back unfiltered
    return copy_through(xyz, normals, n, out_xyz, out_normals, n_out);
  f.mul[0] = 1, f.mul[1] = div_b[0], f.mul[2] = div_b[0] * div_b[1];

  unsigned* d_keys = d.alloc<unsigned>(n);
  unsigned* d_keys_s = d.alloc<unsigned>(n);
  int* d_ids = d.alloc<int>(n);
  int* d_ids_s = d.alloc<int>(n);
  int* d_head = d.alloc<int>(n);
  int* d_slot = d.alloc<int>(n);
  float* d_out = d.alloc<float>(3 * n);
  float* d_out_n = out_normals ? d.alloc<float>(3 * n) : nullptr;
  if (!d_keys || !d_keys_s || !d_ids || !d_ids_s || !d_head || !d_slot || !d_out ||
      (out_normals && !d_out_n))
    return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  hipLaunchKernelGGL(voxel_key_kernel, blocks_for(n), dim3(FILTER_THREADS), 0, st, d_xyz, ni, f, d_keys, d_ids);
  HIP_TRY(ctx, hipGetLastError());
  size_t sort_bytes = 0, scan_bytes = 0;
  HIP_TRY(ctx, rocprim::radix_sort_pairs(nullptr, sort_bytes, d_keys, d_keys_s, d_ids, d_ids_s, n, 0, 32, st));
  HIP_TRY(ctx, rocprim::exclusive_scan(nullptr, scan_bytes, d_head, d_slot, 0, n, rocprim::plus<int>(), st));
  void* d_tmp = d.alloc<char>(std::max(sort_bytes, scan_bytes));
  if (!d_tmp) return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  HIP_TRY(ctx, rocprim::radix_sort_pairs(d_tmp, sort_bytes, d_keys, d_keys_s, d_ids, d_ids_s, n, 0, 32, st));
  hipLaunchKernelGGL(run_head_kernel, blocks_for(n), dim3(FILTER_THREADS), 0, st, d_keys_s, ni, d_head);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, rocprim::exclusive_scan(d_tmp, scan_bytes, d_head, d_slot, 0, n, rocprim::plus<int>(), st));
  hipLaunchKernelGGL(voxel_mean_kernel, blocks_for(n), dim3(FILTER_THREADS), 0, st, d_xyz,
                     (const float*)(out_normals ? d_nrm : nullptr), d_keys_s, d_ids_s, d_head, d_slot,
                     ni, d_out, d_out_n);
  HIP_TRY(ctx, hipGetLastError());
  int last[2] = {0, 0};
  HIP_TRY(ctx, hipMemcpyAsync(&last[0], d_head + (n - 1), sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(&last[1], d_slot + (n - 1), sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  const size_t m = (size_t)last[0] + (size_t)last[1];
  HIP_TRY(ctx, hipMemcpyAsync(out_xyz, d_out, sizeof(float) * 3 * m, hipMemcpyDeviceToHost, st));
  if (out_normals)
    HIP_TRY(ctx, hipMemcpyAsync(out_normals, d_out_n, sizeof(float) * 3 * m, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  *n_out = m;
  return LGICP_OK;
}

int lgicp_random_sample(lgicp_ctx* ctx, const float* xyz, const float* normals, size_t n,
                        size_t n_keep, uint32_t seed, float* out_xyz, float* out_normals,
                        size_t* n_out) {
  if (!ctx || !n_out || (n && (!xyz || !out_xyz)) || (out_normals && !normals))
    return fail(ctx, LGICP_E_INVALID, "null argument");
  if (n > (size_t)1 << 30) return fail(ctx, LGICP_E_INVALID, "cloud too large");
  *n_out = 0;
  if (n == 0 || n_keep == 0) return LGICP_OK;
  if (n_keep >= n) return copy_through(xyz, normals, n, out_xyz, out_normals, n_out);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t st = ctx->stream;
  Dev d;
  float* d_xyz = d.alloc<float>(3 * n);
  float* d_nrm = out_normals ? d.alloc<float>(3 * n) : nullptr;
  unsigned* d_keys = d.alloc<unsigned>(n);
  unsigned* d_keys_s = d.alloc<unsigned>(n);
  int* d_ids = d.alloc<int>(n);
  int* d_ids_s = d.alloc<int>(n);
  unsigned* d_pick = d.alloc<unsigned>(n_keep);
  float* d_out = d.alloc<float>(3 * n_keep);
  if (!d_xyz || (out_normals && !d_nrm) || !d_keys || !d_keys_s || !d_ids || !d_ids_s || !d_pick || !d_out)
    return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  HIP_TRY(ctx, hipMemcpyAsync(d_xyz, xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, st));
  if (out_normals)
    HIP_TRY(ctx, hipMemcpyAsync(d_nrm, normals, sizeof(float) * 3 * n, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(sample_key_kernel, blocks_for(n), dim3(FILTER_THREADS), 0, st, (int)n, seed, d_keys, d_ids);
  HIP_TRY(ctx, hipGetLastError());
  size_t b1 = 0, b2 = 0;
  HIP_TRY(ctx, rocprim::radix_sort_pairs(nullptr, b1, d_keys, d_keys_s, d_ids, d_ids_s, n, 0, 32, st));
  HIP_TRY(ctx, rocprim::radix_sort_keys(nullptr, b2, reinterpret_cast<unsigned*>(d_ids_s), d_pick, n_keep, 0, 32, st));
  void* d_tmp = d.alloc<char>(std::max(b1, b2));
  if (!d_tmp) return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  // the n_keep smallest (hash, index) pairs, then back into input order
  HIP_TRY(ctx, rocprim::radix_sort_pairs(d_tmp, b1, d_keys, d_keys_s, d_ids, d_ids_s, n, 0, 32, st));
  HIP_TRY(ctx, rocprim::radix_sort_keys(d_tmp, b2, reinterpret_cast<unsigned*>(d_ids_s), d_pick, n_keep, 0, 32, st));
  hipLaunchKernelGGL(gather3_kernel, blocks_for(n_keep), dim3(FILTER_THREADS), 0, st, d_xyz, d_pick, (int)n_keep, d_out);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(out_xyz, d_out, sizeof(float) * 3 * n_keep, hipMemcpyDeviceToHost, st));
  if (out_normals) {
    HIP_TRY(ctx, hipStreamSynchronize(st));  // d_out is reused for the normals
    hipLaunchKernelGGL(gather3_kernel, blocks_for(n_keep), dim3(FILTER_THREADS), 0, st, d_nrm, d_pick, (int)n_keep, d_out);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out_normals, d_out, sizeof(float) * 3 * n_keep, hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(ctx, hipStreamSynchronize(st));
  *n_out = n_keep;
  return LGICP_OK;
}

}  // extern "C"
