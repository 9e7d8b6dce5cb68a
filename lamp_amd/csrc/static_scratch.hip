// static_scratch.hip — the one entry point of the C ABI that allocates nothing.
//
// lgicp_nn_search_static is lgicp_nn_search (one pass of the correspondence search of
// lamp_utils/include/lamp_utils/gicp.hpp:463-498) for callers that cannot call hipMalloc: code
// under HIP graph capture, tight serving loops, harnesses that forbid allocation.  Inputs and
// outputs are device buffers of the caller; everything the kernels need beyond them (the padded
// float4 copies of both clouds, the packed 1-NN keys the split workgroups merge into, the cloud /
// pair / state tables and the block map) lives in one file-scope device array of fixed size.
// Every GPU holds its own copy of that array at its own address until the process ends; the
// address is looked up once per device (static_scratch_init, called by lgicp_create with the
// device current) and only that pointer is ever handed to kernels, memsets or copies.
//
// All calls on one GPU share the array, which keeps the previous call's contents: the set-up
// kernel rewrites every region a call reads (tables, block map, both padded clouds up to their
// padded sizes, the keys).  Calls that overlap on one GPU race.
#include <hip/hip_runtime.h>

#include <mutex>

#include "lgicp_internal.h"

namespace {

constexpr size_t kMaxPoints = LGICP_STATIC_MAX_POINTS;  // per cloud
constexpr int kQueriesPerBlock = NN_THREADS * 4;        // nn_kernel<4>
constexpr int kMaxQueryTiles = (int)(kMaxPoints / kQueriesPerBlock);
constexpr int kMaxSplits = 64;
constexpr size_t kMaxBlocks = (size_t)kMaxQueryTiles * kMaxSplits;

// byte offsets inside the scratch array (all multiples of 256)
constexpr size_t kOffClouds = 0;                                  // CloudDev[2]
constexpr size_t kOffPair = 1024;                                 // PairDev
constexpr size_t kOffState = 2048;                                // PairState
constexpr size_t kOffBmap = 4096;                                 // int4[kMaxBlocks]
constexpr size_t kOffKeys = kOffBmap + sizeof(int4) * kMaxBlocks; // nnkey_t[kMaxPoints]
constexpr size_t kOffSrc = kOffKeys + sizeof(nnkey_t) * kMaxPoints;  // float4[kMaxPoints]
constexpr size_t kOffTgt = kOffSrc + sizeof(float4) * kMaxPoints;    // float4[kMaxPoints]
constexpr size_t kScratchBytes = kOffTgt + sizeof(float4) * kMaxPoints;

static_assert(sizeof(CloudDev) * 2 <= 1024 && sizeof(PairDev) <= 1024 && sizeof(PairState) <= 2048,
              "tables outgrew their regions");
static_assert(kMaxPoints % NN_TILE == 0 && kMaxPoints % kQueriesPerBlock == 0, "whole tiles");

// ~2.6 MiB per GPU.  Host code must never pass this name to a kernel: on the host it is a
// stand-in for the symbol, not a device address.
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

constexpr int kMaxDevices = 64;
unsigned char* g_scratch_ptr[kMaxDevices] = {nullptr};
std::mutex g_scratch_mu;

struct Mat16 {
  float v[16];
};

// Rewrites every region of the scratch array that the search reads: tables and block map (first
// workgroup), the padded float4 copies of both clouds and the keys (all workgroups).
__global__ __launch_bounds__(256) void static_nn_setup_kernel(
    unsigned char* __restrict__ base, const float* __restrict__ src, int n_src, int n_pad_s,
    const float* __restrict__ tgt, int n_tgt, int n_pad_t, Mat16 T, int query_tiles, int splits) {
  float4* const s4 = reinterpret_cast<float4*>(base + kOffSrc);
  float4* const t4 = reinterpret_cast<float4*>(base + kOffTgt);
  nnkey_t* const keys = reinterpret_cast<nnkey_t*>(base + kOffKeys);
  const int stride = gridDim.x * blockDim.x;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = gid; i < n_pad_s; i += stride) {
    s4[i] = i < n_src ? make_float4(src[3 * (size_t)i], src[3 * (size_t)i + 1], src[3 * (size_t)i + 2], 1.f)
                      : make_float4(PAD_COORD, PAD_COORD, PAD_COORD, 1.f);
    if (i < n_src) keys[i] = NNKEY_NONE;
  }
  for (int i = gid; i < n_pad_t; i += stride)
    t4[i] = i < n_tgt ? make_float4(tgt[3 * (size_t)i], tgt[3 * (size_t)i + 1], tgt[3 * (size_t)i + 2], 1.f)
                      : make_float4(PAD_COORD, PAD_COORD, PAD_COORD, 1.f);
  if (blockIdx.x != 0) return;
  int4* const bmap = reinterpret_cast<int4*>(base + kOffBmap);
  for (int b = threadIdx.x; b < query_tiles * splits; b += blockDim.x)
    bmap[b] = make_int4(0, b / splits, b % splits, splits);
  if (threadIdx.x == 0) {
    CloudDev* const c = reinterpret_cast<CloudDev*>(base + kOffClouds);
    CloudDev z;
    z.xyz = nullptr, z.nrm = nullptr, z.nrm_x = nullptr, z.cov = nullptr, z.rec = nullptr;
    z.cnrm = nullptr, z.compact = 0;
    z.n = 0, z.n_pad = 0;
    z.sxyz = nullptr, z.tmin = nullptr, z.tmax = nullptr, z.qorder = nullptr;
    z.n_tiles = 0, z.n_q = 0;
    z.centroid[0] = z.centroid[1] = z.centroid[2] = 0.f, z.factor = 0.f;
    c[0] = z, c[1] = z;
    c[0].xyz = s4, c[0].n = n_src, c[0].n_pad = n_pad_s;
    c[1].xyz = t4, c[1].n = n_tgt, c[1].n_pad = n_pad_t;
    PairDev* const p = reinterpret_cast<PairDev*>(base + kOffPair);
    p->src_cloud = 0, p->tgt_cloud = 1, p->scan1_cloud = 0, p->scan2_cloud = 1;
    p->out = s4, p->keys = keys, p->cert = nullptr;
    for (int i = 0; i < 16; ++i) p->guess[i] = (i % 5 == 0) ? 1.f : 0.f;
    p->n_src = n_src, p->n_tgt = n_tgt;
    PairState* const s = reinterpret_cast<PairState*>(base + kOffState);
    for (int i = 0; i < 16; ++i) {
      s->T[i] = T.v[i];
      s->Tprev[i] = s->Tfinal[i] = (i % 5 == 0) ? 1.f : 0.f;
      s->Thash[i] = 0ull;
    }
    s->delta = 0.0, s->iterations = 0, s->converged = 0, s->active = 1, s->failed = 0;
    s->m = 0, s->cycle = 0, s->m_sum = 0.0;
  }
}

inline int round_up(int n, int m) { return ((n + m - 1) / m) * m; }

}  // namespace

namespace lgicp_detail {

// Looks the scratch array's device address up for `device` (which must be the current device).
int static_scratch_init(int device) {
  if (device < 0 || device >= kMaxDevices)
    return fail(nullptr, LGICP_E_INVALID, "device %d beyond the %d the static scratch tracks", device,
                kMaxDevices);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (g_scratch_ptr[device]) return LGICP_OK;
  unsigned char* ptr = nullptr;
  const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
  if (e != hipSuccess || !ptr)
    return fail(nullptr, LGICP_E_HIP, "hipGetSymbolAddress(scratch) failed on device %d: %s", device,
                hipGetErrorString(e));
  g_scratch_ptr[device] = ptr;
  return LGICP_OK;
}

}  // namespace lgicp_detail

extern "C" int lgicp_nn_search_static(lgicp_ctx* ctx, const float T[16], const float* d_src_xyz,
                                      size_t n_src, const float* d_tgt_xyz, size_t n_tgt,
                                      int32_t* d_idx, float* d_d2, void* hip_stream) {
  if (!ctx || !T || (n_src && (!d_src_xyz || !d_idx || !d_d2)) || (n_tgt && !d_tgt_xyz))
    return fail(ctx, LGICP_E_INVALID, "null argument");
  if (n_src > kMaxPoints || n_tgt > kMaxPoints)
    return fail(ctx, LGICP_E_INVALID,
                "lgicp_nn_search_static holds at most %zu points per cloud in its static scratch "
                "(got %zu source, %zu target points): use lgicp_nn_search",
                kMaxPoints, n_src, n_tgt);
  if (n_src == 0) return LGICP_OK;
  const int device = ctx->cfg.device;
  HIP_TRY(ctx, hipSetDevice(device));
  unsigned char* base = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    base = device < kMaxDevices ? g_scratch_ptr[device] : nullptr;
  }
  if (!base) return fail(ctx, LGICP_E_HIP, "static scratch of device %d was never initialised", device);
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  const int ns = (int)n_src, nt = (int)n_tgt;
  const int n_pad_s = std::max(NN_TILE, round_up(ns, NN_TILE));
  const int n_pad_t = std::max(NN_TILE, round_up(nt, NN_TILE));
  const int qt = (ns + kQueriesPerBlock - 1) / kQueriesPerBlock;
  const int tiles = n_pad_t / NN_TILE;
  int sp = 1;
  while ((long long)qt * sp < 1024 && sp < kMaxSplits) sp *= 2;
  sp = std::max(1, std::min(sp, tiles));
  // what the kernels index must fit what the array holds
  if ((size_t)n_pad_s > kMaxPoints || (size_t)n_pad_t > kMaxPoints || (size_t)qt * sp > kMaxBlocks)
    return fail(ctx, LGICP_E_INVALID, "static scratch layout exceeded");
  Mat16 Tm;
  std::memcpy(Tm.v, T, sizeof(Tm.v));
  const int setup_blocks = std::max(1, std::min(1024, (std::max(n_pad_s, n_pad_t) + 255) / 256));
  hipLaunchKernelGGL(static_nn_setup_kernel, dim3(setup_blocks), dim3(256), 0, st, base, d_src_xyz,
                     ns, n_pad_s, d_tgt_xyz, nt, n_pad_t, Tm, qt, sp);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, launch_nn_search(reinterpret_cast<const CloudDev*>(base + kOffClouds),
                                reinterpret_cast<const PairDev*>(base + kOffPair),
                                reinterpret_cast<const PairState*>(base + kOffState),
                                reinterpret_cast<const int4*>(base + kOffBmap), qt * sp,
                                /*final_pass=*/false, /*any_split=*/true, /*variant=*/4, st));
  HIP_TRY(ctx, launch_keys_unpack(reinterpret_cast<const nnkey_t*>(base + kOffKeys), ns, d_idx, d_d2, st));
  if (!hip_stream) HIP_TRY(ctx, hipStreamSynchronize(st));
  return LGICP_OK;
}
