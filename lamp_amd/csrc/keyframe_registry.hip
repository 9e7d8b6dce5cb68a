// keyframe_registry.hip — keyframe registration on the device.
//
// What the reference does per alignment inside its timed region — the kd-tree build of
// setInputTarget (lamp_utils/include/lamp_utils/gicp.h:175-209) and computeCovariances
// (lamp_utils/include/lamp_utils/gicp.hpp:422-432) — happens here once per keyframe, for a whole
// batch of keyframes at a time:
//   * one device allocation and one upload per batch (points and normals as float4 through a
//     pinned staging buffer),
//   * the spatial index: a balanced k-d partition of every keyframe, built level by level for
//     all keyframes together — per level one pass that reduces the box of every range, one pass
//     that keys every point with (range, coordinate along the range's longest axis) and one
//     radix sort over all points of the batch; ranges split at tile boundaries (median split),
//     so after the last level every run of SI_TILE points is a compact tile,
//   * tiles / boxes / query order (index_fill_kernel), covariances from the normals, flagged
//     normals and the 32-byte records in one fused pass, the NormalizePCloud constants of all keyframes in one
//     launch (one wave per keyframe, sequential float sums by contract),
//   * no stream synchronisation: the constants land in pinned host memory and are picked up by
//     the next call that needs them (lgicp_submit, lgicp_wait_registered).
// The shape of the k-d partition is a function of the number of finite points alone, so no range
// tables are needed: every point finds its range at a level by descending from the root.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "lgicp_internal.h"

namespace lgicp {

struct RegKf {  // one keyframe of a registration batch
  const float4* xyz;  // n_pad points
  const float4* nrm;  // n normals
  double* cov;        // n x 6 (null: index only)
  float4* rec;        // n x 2: point, flagged normal (CloudDev::rec, compact form)
  float4* cnrm;       // n: flagged normal
  int* order;         // n: final visiting order (local indices)
  int pos0;           // first global position
  int n;
  int nf;             // finite points
  int tile_base;      // first entry of this keyframe in the per-range box arrays
};

__device__ __forceinline__ unsigned flip_f(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unflip_f(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

__device__ __forceinline__ int kf_of_pos(const RegKf* __restrict__ kfs, int n_kf, int pos) {
  int lo = 0, hi = n_kf - 1;  // last keyframe whose pos0 <= pos
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (kfs[mid].pos0 <= pos) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// The range of local position p at `level`: first position and size; sort = the range is split
// at this level (or, at level 0, has non-finite points to move behind the finite ones).
struct KdRange {
  int lo, m;
  bool sort;
};
__device__ __forceinline__ KdRange kd_range(int p, int n, int nf, int level) {
  KdRange r;
  if (level == 0) {
    r.lo = 0, r.m = n, r.sort = nf > SI_TILE || nf < n;
    return r;
  }
  if (p >= nf) {  // the non-finite tail
    r.lo = nf, r.m = n - nf, r.sort = false;
    return r;
  }
  int lo = 0, m = nf;
  for (int l = 0; l < level && m > SI_TILE; ++l) {
    const int tiles = (m + SI_TILE - 1) / SI_TILE;
    const int k = (tiles / 2) * SI_TILE;  // left half: whole tiles
    if (p < lo + k) m = k; else lo += k, m -= k;
  }
  r.lo = lo, r.m = m, r.sort = m > SI_TILE;
  return r;
}

__global__ __launch_bounds__(256) void kd_init_kernel(const RegKf* __restrict__ kfs, int n_kf, int total,
                                                      int* __restrict__ ids) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= total) return;
  ids[pos] = pos - kfs[kf_of_pos(kfs, n_kf, pos)].pos0;
}

// box of every range that is split at this level
__global__ __launch_bounds__(256) void kd_box_kernel(const RegKf* __restrict__ kfs, int n_kf, int total,
                                                     int level, const int* __restrict__ ids,
                                                     unsigned* __restrict__ bmin,
                                                     unsigned* __restrict__ bmax) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  int slot = -1;
  unsigned kx = 0, ky = 0, kz = 0;
  if (pos < total) {
    const RegKf& K = kfs[kf_of_pos(kfs, n_kf, pos)];
    const int p = pos - K.pos0;
    const KdRange r = kd_range(p, K.n, K.nf, level);
    if (r.sort) {
      const float4 q = K.xyz[ids[pos]];
      if (__builtin_isfinite(q.x) && __builtin_isfinite(q.y) && __builtin_isfinite(q.z)) {
        slot = K.tile_base + r.lo / SI_TILE;
        kx = flip_f(q.x), ky = flip_f(q.y), kz = flip_f(q.z);
      }
    }
  }
  // a wave that lies inside one range (the rule above the last levels) reduces first
  const int first = __builtin_amdgcn_readfirstlane(slot);
  if (__all(slot == first)) {
    if (first < 0) return;
    unsigned x0 = kx, y0 = ky, z0 = kz, x1 = kx, y1 = ky, z1 = kz;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      x0 = min(x0, (unsigned)__shfl_xor((int)x0, off)), x1 = max(x1, (unsigned)__shfl_xor((int)x1, off));
      y0 = min(y0, (unsigned)__shfl_xor((int)y0, off)), y1 = max(y1, (unsigned)__shfl_xor((int)y1, off));
      z0 = min(z0, (unsigned)__shfl_xor((int)z0, off)), z1 = max(z1, (unsigned)__shfl_xor((int)z1, off));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&bmin[3 * first], x0), atomicMax(&bmax[3 * first], x1);
      atomicMin(&bmin[3 * first + 1], y0), atomicMax(&bmax[3 * first + 1], y1);
      atomicMin(&bmin[3 * first + 2], z0), atomicMax(&bmax[3 * first + 2], z1);
    }
    return;
  }
  if (slot < 0) return;
  atomicMin(&bmin[3 * slot], kx), atomicMax(&bmax[3 * slot], kx);
  atomicMin(&bmin[3 * slot + 1], ky), atomicMax(&bmax[3 * slot + 1], ky);
  atomicMin(&bmin[3 * slot + 2], kz), atomicMax(&bmax[3 * slot + 2], kz);
}

// sort key of every point: (first global position of its range, coordinate along the range's
// longest axis); ranges that are not split keep their order (the sort is stable)
__global__ __launch_bounds__(256) void kd_key_kernel(const RegKf* __restrict__ kfs, int n_kf, int total,
                                                     int level, const int* __restrict__ ids,
                                                     const unsigned* __restrict__ bmin,
                                                     const unsigned* __restrict__ bmax,
                                                     unsigned long long* __restrict__ keys) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= total) return;
  const RegKf& K = kfs[kf_of_pos(kfs, n_kf, pos)];
  const int p = pos - K.pos0;
  const KdRange r = kd_range(p, K.n, K.nf, level);
  unsigned low = 0;
  if (r.sort) {
    const float4 q = K.xyz[ids[pos]];
    if (!(__builtin_isfinite(q.x) && __builtin_isfinite(q.y) && __builtin_isfinite(q.z))) {
      low = 0xFFFFFFFFu;  // behind every finite point (only the root range holds such points)
    } else {
      const int slot = K.tile_base + r.lo / SI_TILE;
      const float ex = unflip_f(bmax[3 * slot]) - unflip_f(bmin[3 * slot]);
      const float ey = unflip_f(bmax[3 * slot + 1]) - unflip_f(bmin[3 * slot + 1]);
      const float ez = unflip_f(bmax[3 * slot + 2]) - unflip_f(bmin[3 * slot + 2]);
      int ax = 0;
      float e = ex;
      if (ey > e) ax = 1, e = ey;
      if (ez > e) ax = 2;
      low = flip_f(ax == 0 ? q.x : (ax == 1 ? q.y : q.z));
      if (low == 0xFFFFFFFFu) low = 0xFFFFFFFEu;
    }
  }
  keys[pos] = ((unsigned long long)(unsigned)(K.pos0 + r.lo) << 32) | low;
}

__global__ __launch_bounds__(256) void kd_order_out_kernel(const RegKf* __restrict__ kfs, int n_kf,
                                                           int total, const int* __restrict__ ids) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= total) return;
  const RegKf& K = kfs[kf_of_pos(kfs, n_kf, pos)];
  K.order[pos - K.pos0] = ids[pos];
}

// covariance from the stored normal (gicp_utils.h:28-62, all in double; arithmetic identical to
// cov_from_normals_kernel), the flagged normal and the 32-byte record (point + flagged normal) the
// reduction reads instead of the covariance wherever the closed form holds, in one pass
__global__ __launch_bounds__(256) void derive_kernel(const RegKf* __restrict__ kfs) {
  const RegKf& K = kfs[blockIdx.y];
  if (!K.cov) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < K.n; i += gridDim.x * blockDim.x) {
    const float4 nf = K.nrm[i];
    double nx = nf.x, ny = nf.y, nz = nf.z;
    const double z = nx * nx + ny * ny + nz * nz;
    if (z > 0) {
      const double s = sqrt(z);
      nx /= s, ny /= s, nz /= s;
    }
    if (fabsf(nf.z) < 0.0000001 || fabs(nz) < 0.0000001) nz = 0.0000001;
    double v1x = 1.0, v1y = 0.0, v1z = -nx / nz;
    {
      const double s = sqrt(v1x * v1x + v1y * v1y + v1z * v1z);
      v1x /= s, v1y /= s, v1z /= s;
    }
    double v2x = ny * v1z - nz * v1y;
    double v2y = nz * v1x - nx * v1z;
    double v2z = nx * v1y - ny * v1x;
    {
      const double z2 = v2x * v2x + v2y * v2y + v2z * v2z;
      if (z2 > 0) {
        const double s = sqrt(z2);
        v2x /= s, v2y /= s, v2z /= s;
      }
    }
    double c[6];
    c[0] = 0.001 * nx * nx + v1x * v1x + v2x * v2x;
    c[1] = 0.001 * nx * ny + v1x * v1y + v2x * v2y;
    c[2] = 0.001 * nx * nz + v1x * v1z + v2x * v2z;
    c[3] = 0.001 * ny * ny + v1y * v1y + v2y * v2y;
    c[4] = 0.001 * ny * nz + v1y * v1z + v2y * v2z;
    c[5] = 0.001 * nz * nz + v1z * v1z + v2z * v2z;
    double2* co = reinterpret_cast<double2*>(K.cov + 6 * (size_t)i);
    co[0] = make_double2(c[0], c[1]), co[1] = make_double2(c[2], c[3]), co[2] = make_double2(c[4], c[5]);
    const float4 cn = make_float4(nf.x, nf.y, nf.z, normal_is_special(nf));
    K.cnrm[i] = cn;
    float4* r = K.rec + 2 * (size_t)i;
    r[0] = K.xyz[i];
    r[1] = cn;
  }
}

}  // namespace lgicp

namespace lgicp_detail {

namespace {
inline int round_up(int n, int m) { return ((n + m - 1) / m) * m; }
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// levels of the k-d partition of m finite points (the right child is never the smaller one)
int kd_depth(int m) {
  int d = 0;
  while (m > SI_TILE) {
    const int tiles = (m + SI_TILE - 1) / SI_TILE;
    m -= (tiles / 2) * SI_TILE;
    ++d;
  }
  return d;
}

hipError_t pinned_reserve(char** p, size_t* cap, size_t want) {
  if (*cap >= want) return hipSuccess;
  if (*p) (void)hipHostFree(*p);
  *p = nullptr, *cap = 0;
  const size_t take = want + want / 4 + 4096;
  const hipError_t e = hipHostMalloc((void**)p, take, hipHostMallocDefault);
  if (e == hipSuccess) *cap = take;
  return e;
}
}  // namespace

void Registry::destroy() {
  if (h_stage) (void)hipHostFree(h_stage);
  if (h_tab) (void)hipHostFree(h_tab);
  if (h_const) (void)hipHostFree(h_const);
  h_stage = h_tab = nullptr, h_const = nullptr;
  stage_cap = tab_cap = const_cap = 0;
  if (ev_stage) (void)hipEventDestroy(ev_stage);
  ev_stage = nullptr;
  d_keys[0].release(), d_keys[1].release(), d_ids[0].release(), d_ids[1].release();
  d_box.release(), d_tmp.release(), d_tab.release();
  pending.clear();
}

// Picks the NormalizePCloud constants of freshly registered keyframes up from pinned memory.
int flush_registrations(lgicp_ctx* ctx) {
  Registry& R = ctx->reg;
  if (R.pending.empty()) return LGICP_OK;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (const Registry::Pending& p : R.pending) {
    auto it = ctx->scans.find(p.key);
    if (it == ctx->scans.end()) continue;
    const CloudDev& c = reinterpret_cast<const CloudDev*>(R.h_const)[p.slot];
    std::memcpy(it->second.centroid, c.centroid, sizeof(c.centroid));
    it->second.factor = c.factor;
  }
  R.pending.clear();
  R.const_used = 0;
  return LGICP_OK;
}

// The k-d visiting order of every keyframe of the table (level by level, all keyframes together),
// tiles / boxes / query order, and (derive) covariances, flagged normals and records.  The tables
// are already on the device; `total` = points of all keyframes, tile_base = their range slots.
int index_and_derive(lgicp_ctx* ctx, Registry& R, hipStream_t st, const lgicp::RegKf* d_kfs,
                     const IndexSegDev* d_segs, int n_kf, int total, int tile_base, int depth,
                     bool any_sort, int max_n, bool derive) {
  // ---- k-d partition, level by level, all keyframes together -----------------------------------
  if (total > 0) {
    const int levels = std::max(depth, any_sort ? 1 : 0);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    for (int k = 0; k < 2; ++k) {
      HIP_TRY(ctx, R.d_keys[k].reserve((size_t)total));
      HIP_TRY(ctx, R.d_ids[k].reserve((size_t)total));
    }
    HIP_TRY(ctx, R.d_box.reserve((size_t)6 * tile_base));
    unsigned* const bmin = R.d_box.p;
    unsigned* const bmax = R.d_box.p + (size_t)3 * tile_base;
    int end_bit = 33;
    while (end_bit < 64 && ((unsigned long long)total >> (end_bit - 32)) != 0) ++end_bit;
    size_t tmp_bytes = 0;
    HIP_TRY(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, R.d_keys[0].p, R.d_keys[1].p,
                                           R.d_ids[0].p, R.d_ids[1].p, (size_t)total, 0, end_bit, st));
    HIP_TRY(ctx, R.d_tmp.reserve(tmp_bytes));
    int cur = 0;
    hipLaunchKernelGGL(lgicp::kd_init_kernel, dim3(blocks), dim3(256), 0, st, d_kfs, n_kf, total,
                       R.d_ids[cur].p);
    for (int level = 0; level < levels; ++level) {
      HIP_TRY(ctx, hipMemsetAsync(bmin, 0xFF, sizeof(unsigned) * 3 * (size_t)tile_base, st));
      HIP_TRY(ctx, hipMemsetAsync(bmax, 0x00, sizeof(unsigned) * 3 * (size_t)tile_base, st));
      hipLaunchKernelGGL(lgicp::kd_box_kernel, dim3(blocks), dim3(256), 0, st, d_kfs, n_kf, total, level,
                         R.d_ids[cur].p, bmin, bmax);
      hipLaunchKernelGGL(lgicp::kd_key_kernel, dim3(blocks), dim3(256), 0, st, d_kfs, n_kf, total, level,
                         R.d_ids[cur].p, bmin, bmax, R.d_keys[0].p);
      HIP_TRY(ctx, hipGetLastError());
      HIP_TRY(ctx, rocprim::radix_sort_pairs(R.d_tmp.p, tmp_bytes, R.d_keys[0].p, R.d_keys[1].p,
                                             R.d_ids[cur].p, R.d_ids[cur ^ 1].p, (size_t)total, 0,
                                             end_bit, st));
      cur ^= 1;
    }
    hipLaunchKernelGGL(lgicp::kd_order_out_kernel, dim3(blocks), dim3(256), 0, st, d_kfs, n_kf, total,
                       R.d_ids[cur].p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, launch_index_fill(d_segs, n_kf, max_n, st));
  }
  if (!derive || total == 0) return LGICP_OK;

  // ---- what depends on a keyframe alone ----------------------------------------------------------
  {
    int bx = std::max(1, std::min(64, (max_n + 255) / 256));
    hipLaunchKernelGGL(lgicp::derive_kernel, dim3(bx, n_kf), dim3(256), 0, st, d_kfs);
    HIP_TRY(ctx, hipGetLastError());
  }
  return LGICP_OK;
}

// Registers `count` clouds: upload, spatial index and (derive) what depends on a keyframe alone.
// out[i] receives the device view of in[i]; the keyframes of one call share one device block.
// Everything is enqueued on ctx->stream; nothing is waited for, except that the staging buffer
// of the previous call must have been consumed.
int register_keyframes(lgicp_ctx* ctx, const KfInput* in, size_t count, Keyframe* out, bool derive) {
  if (count == 0) return LGICP_OK;
  Registry& R = ctx->reg;
  hipStream_t st = ctx->stream;
  const int n_kf = (int)count;

  // ---- layout: [uploaded: xyz, nrm of every keyframe] [derived arrays of every keyframe] --------
  struct Lay {
    size_t xyz, nrm, order, sxyz, tmin, tmax, qorder, cov, rec, cnrm, nrm_x;
    int n, n_pad, n_tiles, n_q, nf;
    bool missing;
  };
  std::vector<Lay> lay(count);
  size_t up_bytes = 0, total_pts = 0;
  for (size_t k = 0; k < count; ++k) {
    Lay& L = lay[k];
    if (in[k].n > (size_t)1 << 28) return fail(ctx, LGICP_E_INVALID, "cloud too large");
    L.n = (int)in[k].n;
    L.n_pad = std::max(NN_TILE, round_up(L.n, NN_TILE));
    L.n_tiles = (L.n + SI_TILE - 1) / SI_TILE;
    L.n_q = round_up(L.n, 64);
    L.xyz = up_bytes, up_bytes += align256(sizeof(float4) * (size_t)L.n_pad);
    L.nrm = up_bytes, up_bytes += align256(sizeof(float4) * (size_t)std::max(L.n, 1));
    total_pts += (size_t)L.n;
  }
  if (total_pts > (size_t)1 << 30) return fail(ctx, LGICP_E_INVALID, "registration batch too large");
  size_t bytes = up_bytes;
  for (size_t k = 0; k < count; ++k) {
    Lay& L = lay[k];
    const size_t n = (size_t)std::max(L.n, 1), t = (size_t)std::max(L.n_tiles, 1);
    L.order = bytes, bytes += align256(sizeof(int) * n);
    L.sxyz = bytes, bytes += align256(sizeof(float4) * t * SI_TILE);
    L.tmin = bytes, bytes += align256(sizeof(float4) * (t + (t + 63) / 64));  // tiles, blocks of 64 tiles
    L.tmax = bytes, bytes += align256(sizeof(float4) * (t + (t + 63) / 64));
    L.qorder = bytes, bytes += align256(sizeof(int) * (size_t)std::max(L.n_q, 64));
    L.cov = L.rec = L.cnrm = L.nrm_x = 0;
    if (derive) {
      L.cov = bytes, bytes += align256(sizeof(double) * 6 * n);
      L.rec = bytes, bytes += align256(sizeof(float4) * 2 * n);
      L.cnrm = bytes, bytes += align256(sizeof(float4) * n);
    }
  }

  // ---- staging: the previous call's upload must have left the pinned buffer ----------------
  if (R.ev_stage) HIP_TRY(ctx, hipEventSynchronize(R.ev_stage));
  else HIP_TRY(ctx, hipEventCreateWithFlags(&R.ev_stage, hipEventDisableTiming));
  HIP_TRY(ctx, pinned_reserve(&R.h_stage, &R.stage_cap, up_bytes));
  // Staging is a plain per-point copy into float4 layout (plus the count of finite points): the
  // host's share of a registration, and for a batch of a million points most of its wall time when
  // one thread does it.  Keyframes are dealt out to a few threads.
  auto stage_one = [&](size_t k) {
    Lay& L = lay[k];
    float4* hx = reinterpret_cast<float4*>(R.h_stage + L.xyz);
    float4* hn = reinterpret_cast<float4*>(R.h_stage + L.nrm);
    const size_t sf = in[k].stride_floats;
    const float* xyz = in[k].xyz;
    const float* nrm = in[k].nrm;
    int nf = 0;
    for (int i = 0; i < L.n; ++i) {
      const float x = xyz[i * sf], y = xyz[i * sf + 1], z = xyz[i * sf + 2];
      hx[i] = make_float4(x, y, z, 1.0f);
      nf += std::isfinite(x) && std::isfinite(y) && std::isfinite(z);
    }
    for (int i = L.n; i < L.n_pad; ++i) hx[i] = make_float4(PAD_COORD, PAD_COORD, PAD_COORD, 1.0f);
    if (nrm)
      for (int i = 0; i < L.n; ++i) hn[i] = make_float4(nrm[i * sf], nrm[i * sf + 1], nrm[i * sf + 2], 0.f);
    else
      std::memset(hn, 0, sizeof(float4) * (size_t)std::max(L.n, 1));
    L.nf = nf;
    // lamp_utils::ExtractNormals looks at the first stored normal only (PointCloudUtils.cc:17-35)
    L.missing = derive && L.n > 0 && hn[0].x == 0.f && hn[0].y == 0.f && hn[0].z == 0.f;
  };
  {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t n_thr = std::min<size_t>({(size_t)8, (size_t)hw, count, total_pts / 65536 + 1});
    if (n_thr <= 1) {
      for (size_t k = 0; k < count; ++k) stage_one(k);
    } else {
      std::atomic<size_t> next{0};
      auto work = [&] {
        for (size_t k = next.fetch_add(1); k < count; k = next.fetch_add(1)) stage_one(k);
      };
      std::vector<std::thread> pool;
      for (size_t t = 1; t < n_thr; ++t) pool.emplace_back(work);
      work();
      for (std::thread& t : pool) t.join();
    }
  }
  for (size_t k = 0; k < count; ++k)
    if (lay[k].missing) lay[k].nrm_x = bytes, bytes += align256(sizeof(float4) * (size_t)lay[k].n);

  auto block = std::make_shared<DevBlock>();
  HIP_TRY(ctx, hipMalloc((void**)&block->p, std::max<size_t>(bytes, 256)));
  char* const base = block->p;
  HIP_TRY(ctx, launch_copy16(base, R.h_stage, up_bytes, st));

  // ---- tables ----------------------------------------------------------------------------------
  // [RegKf x n_kf][IndexSegDev x n_kf] in pinned memory, copied to the device by a kernel; the
  // table of a call is consumed before the next call rewrites it (same stream, same event)
  const size_t tab_bytes = align256(sizeof(RegKf) * count) + align256(sizeof(IndexSegDev) * count);
  HIP_TRY(ctx, pinned_reserve(&R.h_tab, &R.tab_cap, tab_bytes));
  HIP_TRY(ctx, R.d_tab.reserve(tab_bytes));
  RegKf* h_kfs = reinterpret_cast<RegKf*>(R.h_tab);
  IndexSegDev* h_segs = reinterpret_cast<IndexSegDev*>(R.h_tab + align256(sizeof(RegKf) * count));
  int pos = 0, tile_base = 0, depth = 0, max_n = 0;
  bool any_sort = false;
  for (size_t k = 0; k < count; ++k) {
    const Lay& L = lay[k];
    Keyframe& kf = out[k];
    kf = Keyframe();
    kf.block = block;
    kf.n = L.n, kf.n_pad = L.n_pad, kf.n_tiles = L.n_tiles, kf.n_q = L.n_q, kf.nf = L.nf;
    kf.d_xyz = reinterpret_cast<float4*>(base + L.xyz);
    kf.d_nrm = reinterpret_cast<float4*>(base + L.nrm);
    kf.d_order = reinterpret_cast<int*>(base + L.order);
    kf.d_sxyz = reinterpret_cast<float4*>(base + L.sxyz);
    kf.d_tmin = reinterpret_cast<float4*>(base + L.tmin);
    kf.d_tmax = reinterpret_cast<float4*>(base + L.tmax);
    kf.d_qorder = reinterpret_cast<int*>(base + L.qorder);
    kf.normals_missing = L.missing;
    if (derive) {
      kf.d_cov = reinterpret_cast<double*>(base + L.cov);
      kf.d_rec = reinterpret_cast<float4*>(base + L.rec);
      kf.d_cnrm = reinterpret_cast<float4*>(base + L.cnrm);
      kf.d_nrm_x = L.missing ? reinterpret_cast<float4*>(base + L.nrm_x) : kf.d_nrm;
    }
    RegKf& g = h_kfs[k];
    g.xyz = kf.d_xyz, g.nrm = kf.d_nrm, g.cov = kf.d_cov, g.rec = kf.d_rec, g.cnrm = kf.d_cnrm, g.order = kf.d_order;
    g.pos0 = pos, g.n = L.n, g.nf = L.nf, g.tile_base = tile_base;
    pos += L.n;
    tile_base += L.n_tiles + 1;
    depth = std::max(depth, kd_depth(L.nf));
    any_sort = any_sort || L.nf < L.n;
    max_n = std::max(max_n, L.n);
    IndexSegDev& s = h_segs[k];
    std::memset(&s, 0, sizeof(s));
    s.xyz = kf.d_xyz, s.order = kf.d_order, s.sxyz = kf.d_sxyz, s.tmin = kf.d_tmin, s.tmax = kf.d_tmax;
    s.qorder = kf.d_qorder, s.point_off = 0, s.tile_off = 0, s.n = L.n, s.last = 1, s.n_q = L.n_q;
  }
  const int total = pos;
  const RegKf* d_kfs = reinterpret_cast<const RegKf*>(R.d_tab.p);
  const IndexSegDev* d_segs =
      reinterpret_cast<const IndexSegDev*>(R.d_tab.p + align256(sizeof(RegKf) * count));
  HIP_TRY(ctx, launch_copy16(R.d_tab.p, R.h_tab, tab_bytes, st));
  // both pinned buffers (points, tables) are free for the next call once this event has passed
  HIP_TRY(ctx, hipEventRecord(R.ev_stage, st));

  {
    const int rc = index_and_derive(ctx, R, st, d_kfs, d_segs, n_kf, total, tile_base, depth, any_sort, max_n, derive);
    if (rc != LGICP_OK) return rc;
  }
  if (!derive || total == 0) return LGICP_OK;
  for (size_t k = 0; k < count; ++k)  // ExtractNormals falls back to ComputeNormals (k = 10)
    if (lay[k].missing)
      HIP_TRY(ctx, launch_normals_knn(out[k].d_xyz, out[k].n, out[k].n_pad, std::min(10, out[k].n),
                                      out[k].d_nrm_x, st));
  // NormalizePCloud constants: CloudDev entries in pinned, device-visible memory; the kernel
  // writes centroid / factor there and flush_registrations() reads them after the next sync
  const size_t need = sizeof(CloudDev) * (R.const_used + count) + sizeof(int) * (R.const_used + count);
  if (need > R.const_cap) {
    // growing moves the buffer: results still pending in the old one are fetched first
    const int rc = flush_registrations(ctx);
    if (rc != LGICP_OK) return rc;
    char* p = reinterpret_cast<char*>(R.h_const);
    size_t cap = R.const_cap;
    HIP_TRY(ctx, pinned_reserve(&p, &cap, std::max<size_t>(need, (size_t)64 << 10)));
    R.h_const = p, R.const_cap = cap;
  }
  const size_t max_slots = R.const_cap / (sizeof(CloudDev) + sizeof(int));
  CloudDev* h_clouds = reinterpret_cast<CloudDev*>(R.h_const);
  int* h_ids = reinterpret_cast<int*>(R.h_const + sizeof(CloudDev) * max_slots);
  const size_t first = R.const_used;
  for (size_t k = 0; k < count; ++k) {
    CloudDev& c = h_clouds[first + k];
    std::memset(&c, 0, sizeof(c));
    c.xyz = out[k].d_xyz, c.n = out[k].n, c.n_pad = out[k].n_pad;
    h_ids[first + k] = (int)(first + k);
    R.pending.push_back({in[k].key, first + k});
  }
  R.const_used += count;
  CloudDev* d_clouds = nullptr;
  int* d_ids = nullptr;
  HIP_TRY(ctx, hipHostGetDevicePointer((void**)&d_clouds, h_clouds, 0));
  HIP_TRY(ctx, hipHostGetDevicePointer((void**)&d_ids, h_ids, 0));
  HIP_TRY(ctx, launch_normalize_constants(d_clouds, d_ids + first, (int)count, st));
  return LGICP_OK;
}

// Accumulated clouds as keyframes of their own (see AccEntry): points and normals come from the
// resident member keyframes through assemble_kernel instead of an upload; everything after that is
// what registration does.
int register_accumulated(lgicp_ctx* ctx, Registry& R, hipStream_t st, AccBuild* builds, size_t count) {
  if (count == 0) return LGICP_OK;
  const int n_kf = (int)count;
  size_t n_segs = 0;
  for (size_t k = 0; k < count; ++k) n_segs += 1 + builds[k].segs.size();
  // tables: [RegKf x n_kf][IndexSegDev x n_kf][SegmentDev x n_segs]
  const size_t o_iseg = align256(sizeof(lgicp::RegKf) * count);
  const size_t o_seg = o_iseg + align256(sizeof(IndexSegDev) * count);
  const size_t tab_bytes = o_seg + align256(sizeof(SegmentDev) * n_segs);
  if (R.ev_stage) HIP_TRY(ctx, hipEventSynchronize(R.ev_stage));
  else HIP_TRY(ctx, hipEventCreateWithFlags(&R.ev_stage, hipEventDisableTiming));
  HIP_TRY(ctx, pinned_reserve(&R.h_tab, &R.tab_cap, tab_bytes));
  HIP_TRY(ctx, R.d_tab.reserve(tab_bytes));
  lgicp::RegKf* h_kfs = reinterpret_cast<lgicp::RegKf*>(R.h_tab);
  IndexSegDev* h_isegs = reinterpret_cast<IndexSegDev*>(R.h_tab + o_iseg);
  SegmentDev* h_segs = reinterpret_cast<SegmentDev*>(R.h_tab + o_seg);
  int pos = 0, tile_base = 0, depth = 0, max_n = 0, max_seg_n = 0;
  size_t seg_i = 0;
  for (size_t k = 0; k < count; ++k) {
    AccBuild& B = builds[k];
    int n = B.own->n, nf = B.own->nf;
    for (auto& sg : B.segs) n += sg.first->n, nf += sg.first->nf;
    if (nf != n) return fail(ctx, LGICP_E_INVALID, "accumulated cloud with non-finite points");
    const int n_pad = std::max(NN_TILE, round_up(n, NN_TILE));
    const int n_tiles = (n + SI_TILE - 1) / SI_TILE, n_q = round_up(n, 64);
    const size_t np = (size_t)std::max(n, 1), t = (size_t)std::max(n_tiles, 1);
    size_t bytes = 0;
    auto take = [&](size_t b) {
      const size_t o = bytes;
      bytes += align256(b);
      return o;
    };
    const size_t o_xyz = take(sizeof(float4) * (size_t)n_pad), o_nrm = take(sizeof(float4) * np);
    const size_t o_order = take(sizeof(int) * np), o_sxyz = take(sizeof(float4) * t * SI_TILE);
    const size_t o_tmin = take(sizeof(float4) * (t + (t + 63) / 64));
    const size_t o_tmax = take(sizeof(float4) * (t + (t + 63) / 64));
    const size_t o_qorder = take(sizeof(int) * (size_t)std::max(n_q, 64));
    const size_t o_cov = take(sizeof(double) * 6 * np), o_rec = take(sizeof(float4) * 2 * np);
    const size_t o_cnrm = take(sizeof(float4) * np);
    // freed through the parking list: hipFree would wait for every stream of the device
    std::shared_ptr<DevBlock> block(new DevBlock, [](DevBlock* b) {
      park_device_memory(b->p);
      b->p = nullptr;
      delete b;
    });
    HIP_TRY(ctx, hipMalloc((void**)&block->p, std::max<size_t>(bytes, 256)));
    char* const base = block->p;
    Keyframe& kf = B.out->kf;
    kf = Keyframe();
    kf.block = block;
    kf.n = n, kf.n_pad = n_pad, kf.n_tiles = n_tiles, kf.n_q = n_q, kf.nf = nf;
    kf.d_xyz = reinterpret_cast<float4*>(base + o_xyz);
    kf.d_nrm = reinterpret_cast<float4*>(base + o_nrm);
    kf.d_nrm_x = kf.d_nrm;
    kf.d_order = reinterpret_cast<int*>(base + o_order);
    kf.d_sxyz = reinterpret_cast<float4*>(base + o_sxyz);
    kf.d_tmin = reinterpret_cast<float4*>(base + o_tmin);
    kf.d_tmax = reinterpret_cast<float4*>(base + o_tmax);
    kf.d_qorder = reinterpret_cast<int*>(base + o_qorder);
    kf.d_cov = reinterpret_cast<double*>(base + o_cov);
    kf.d_rec = reinterpret_cast<float4*>(base + o_rec);
    kf.d_cnrm = reinterpret_cast<float4*>(base + o_cnrm);
    B.out->bytes = bytes;
    lgicp::RegKf& g = h_kfs[k];
    g.xyz = kf.d_xyz, g.nrm = kf.d_nrm, g.cov = kf.d_cov, g.rec = kf.d_rec, g.cnrm = kf.d_cnrm, g.order = kf.d_order;
    g.pos0 = pos, g.n = n, g.nf = nf, g.tile_base = tile_base;
    pos += n;
    tile_base += n_tiles + 1;
    depth = std::max(depth, kd_depth(nf));
    max_n = std::max(max_n, n);
    IndexSegDev& is = h_isegs[k];
    std::memset(&is, 0, sizeof(is));
    is.xyz = kf.d_xyz, is.order = kf.d_order, is.sxyz = kf.d_sxyz, is.tmin = kf.d_tmin, is.tmax = kf.d_tmax;
    is.qorder = kf.d_qorder, is.point_off = 0, is.tile_off = 0, is.n = n, is.last = 1, is.n_q = n_q;
    // AccumulateScans: the key's own scan, then every neighbour moved into its frame
    int off = 0;
    auto seg = [&](const Keyframe* src, const hm::Pose* T) {
      SegmentDev& sd = h_segs[seg_i++];
      std::memset(&sd, 0, sizeof(sd));
      sd.src_xyz = src->d_xyz, sd.src_nrm = src->d_nrm;
      sd.dst_xyz = kf.d_xyz + off, sd.dst_nrm = kf.d_nrm + off;
      sd.n = src->n;
      sd.identity = T ? 0 : 1;
      if (T)
        for (int a = 0; a < 3; ++a) {
          for (int b = 0; b < 3; ++b) sd.T[a * 4 + b] = T->R[a * 3 + b];
          sd.T[a * 4 + 3] = T->t[a];
        }
      max_seg_n = std::max(max_seg_n, src->n);
      off += src->n;
    };
    seg(B.own, nullptr);
    for (auto& sg : B.segs) seg(sg.first, &sg.second);
  }
  const int total = pos;
  HIP_TRY(ctx, launch_copy16(R.d_tab.p, R.h_tab, tab_bytes, st));
  HIP_TRY(ctx, hipEventRecord(R.ev_stage, st));
  const lgicp::RegKf* d_kfs = reinterpret_cast<const lgicp::RegKf*>(R.d_tab.p);
  const IndexSegDev* d_isegs = reinterpret_cast<const IndexSegDev*>(R.d_tab.p + o_iseg);
  const SegmentDev* d_segs = reinterpret_cast<const SegmentDev*>(R.d_tab.p + o_seg);
  HIP_TRY(ctx, launch_assemble(d_segs, (int)n_segs, max_seg_n, st));
  for (size_t k = 0; k < count; ++k)
    HIP_TRY(ctx, launch_pad_fill(builds[k].out->kf.d_xyz, builds[k].out->kf.n, builds[k].out->kf.n_pad, st));
  return index_and_derive(ctx, R, st, d_kfs, d_isegs, n_kf, total, tile_base, depth, false, max_n, true);
}

}  // namespace lgicp_detail
