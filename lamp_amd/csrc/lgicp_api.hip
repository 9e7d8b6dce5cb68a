// lgicp_api.hip — host driver behind the C ABI of include/lamp_gicp.h.
//
// Mirrors the batch body of IcpLoopComputation::ComputeTransforms / PerformAlignment
// (loop_closure/src/IcpLoopComputation.cc:243-367, 420-648): keyframe and pose registries
// (first insert wins), per-candidate validity checks, scan accumulation, initial guess, the
// GICP outer loop, fitness / transform gating, PoseInverse and the ICP covariance.  All
// per-point work is enqueued as HIP kernels on one stream; all pairs of a batch advance
// through the outer loop together and drop out individually as they converge.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/lamp_gicp.h"
#include "host_math.h"
#include "lgicp_device.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <limits>
#include <thread>

#include "lgicp_internal.h"

namespace lgicp_detail {

static std::mutex g_parked_mutex;
static std::vector<void*> g_parked, g_parked_pinned;
thread_local lgicp_ctx* t_park_ctx = nullptr;

void park_device_memory(void* p) {
  if (!p) return;
  if (lgicp_ctx* c = t_park_ctx) {
    std::lock_guard<std::mutex> lk(c->parked_mu);
    c->parked_dev.push_back(p);
    return;
  }
  std::lock_guard<std::mutex> lk(g_parked_mutex);
  g_parked.push_back(p);
}

void park_pinned_memory(void* p) {
  if (!p) return;
  if (lgicp_ctx* c = t_park_ctx) {
    std::lock_guard<std::mutex> lk(c->parked_mu);
    c->parked_pinned.push_back(p);
    return;
  }
  std::lock_guard<std::mutex> lk(g_parked_mutex);
  g_parked_pinned.push_back(p);
}

void free_parked_device_memory(lgicp_ctx* ctx) {
  std::vector<void*> dev, pinned;
  if (ctx) {
    std::lock_guard<std::mutex> lk(ctx->parked_mu);
    dev.swap(ctx->parked_dev);
    pinned.swap(ctx->parked_pinned);
  } else {
    std::lock_guard<std::mutex> lk(g_parked_mutex);
    dev.swap(g_parked);
    pinned.swap(g_parked_pinned);
  }
  for (void* p : dev) (void)hipFree(p);
  for (void* p : pinned) (void)hipHostFree(p);
}

}  // namespace lgicp_detail

namespace {

inline int round_up(int n, int m) { return ((n + m - 1) / m) * m; }

struct Arena {
  BatchSlot* slot;
  size_t off = 0;
  template <typename T>
  T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = reinterpret_cast<T*>(slot->arena.p + off);
    off += n * sizeof(T);
    return p;
  }
};

// per-pair device scratch of a batch: transformed source, packed matches, certificates (+ the
// BFGS back end's records, 72 bytes per source point)
inline size_t pair_scratch_bytes(const lgicp_config& cfg, int n_src) {
  return (size_t)n_src * (cfg.solver == LGICP_SOLVER_BFGS ? 40 + 72 : 40) + 1536;
}

struct CloudPlan {  // one distinct point cloud of a batch
  uint64_t key;
  bool accumulated;
  std::vector<std::pair<uint64_t, hm::Pose>> segs;  // appended neighbour keyframes + tf
  int n = 0, n_pad = 0;
  bool need_norm_constants = false;
  // accumulated and kept on the device from submit to submit (AccEntry): looked up or built when
  // the batch starts; such a cloud then reads like a registered keyframe
  bool cacheable = false;
  std::shared_ptr<AccEntry> entry;
  // in the arena of the batch (accumulated and not cached)
  bool in_arena() const { return accumulated && !entry; }
};

// XCD-aware order of workgroup ids: the dispatcher deals consecutive ids round-robin over the
// 8 XCDs, so logical block (b % 8) * ceil(n/8) + b / 8 keeps neighbouring logical blocks (the
// query tiles of one pair, which stream the same target) on one XCD's L2.
template <typename T>
void xcd_permute(std::vector<T>& v, std::vector<T>& out) {
  const size_t n = v.size();
  if (n < 16) return;
  out.resize(n);
  const size_t q = n / 8, r = n % 8;
  for (size_t b = 0; b < n; ++b) {
    const size_t x = b % 8, k = b / 8;
    const size_t logical = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
    out[b] = v[logical];
  }
  v.swap(out);
}

}  // namespace

namespace {

// Streams whose kernels really run side by side.  The runtime deals its few hardware queues
// (four per process by default) out to streams in creation order, counting the streams the host
// application created before: four fresh streams may well land on three queues, and two parts
// of a submit would then take turns instead of overlapping (measured: 5.8 ms instead of 4.4 ms
// per C2 submit inside a process that had merely imported another GPU library first).  So
// streams are created one by one and a new one is kept only if an idle wave on it overlaps an
// idle wave on every stream kept so far.
// probe = false: `wanted` streams as the runtime hands them out (an explicit request, e.g. under a
// profiler that serialises all kernels, where nothing overlaps and the test would keep one stream).
std::vector<hipStream_t> concurrent_streams(int wanted, bool probe) {
  std::vector<hipStream_t> kept, spare;
  auto fresh = []() -> hipStream_t {
    hipStream_t s = nullptr;
    return hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess ? s : nullptr;
  };
  hipStream_t first = fresh();
  if (!first) return kept;
  kept.push_back(first);
  if (wanted <= 1) return kept;
  if (!probe) {
    while ((int)kept.size() < wanted) {
      hipStream_t s = fresh();
      if (!s) break;
      kept.push_back(s);
    }
    return kept;
  }
  constexpr double kDelayUs = 200.0;
  (void)launch_delay(1.0, first);  // loads the kernel
  (void)hipStreamSynchronize(first);
  hipEvent_t e0 = nullptr, a_end = nullptr, b_start = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&a_end) != hipSuccess ||
      hipEventCreate(&b_start) != hipSuccess)
    return kept;
  // Device timestamps, not host clocks: on a shared queue the marker that opens b's work waits
  // behind a's idle wave, on separate queues it passes at once.  A host thread that loses the CPU
  // between the two launches can only make separate queues look shared, so a few attempts are
  // made and one overlap settles it.
  auto overlap = [&](hipStream_t a, hipStream_t b) {
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipStreamSynchronize(a);
      (void)hipStreamSynchronize(b);
      (void)hipEventRecord(e0, a);
      (void)launch_delay(kDelayUs, a);
      (void)hipEventRecord(a_end, a);
      (void)hipEventRecord(b_start, b);
      (void)launch_delay(kDelayUs, b);
      (void)hipStreamSynchronize(a);
      (void)hipStreamSynchronize(b);
      float t_a_end = 0.f, t_b_start = 0.f;
      if (hipEventElapsedTime(&t_a_end, e0, a_end) != hipSuccess ||
          hipEventElapsedTime(&t_b_start, e0, b_start) != hipSuccess)
        continue;
      if (t_b_start < 0.5f * t_a_end) return true;
    }
    return false;
  };
  for (int tries = 0; tries < 4 * wanted && (int)kept.size() < wanted; ++tries) {
    hipStream_t s = fresh();
    if (!s) break;
    bool ok = true;
    for (hipStream_t k : kept) ok = ok && overlap(s, k);
    (ok ? kept : spare).push_back(s);
  }
  for (hipStream_t s : spare) (void)hipStreamDestroy(s);
  (void)hipEventDestroy(e0), (void)hipEventDestroy(a_end), (void)hipEventDestroy(b_start);
  if (std::getenv("LGICP_TRACE"))
    fprintf(stderr, "[lgicp] streams: %zu kept of %zu created\n", kept.size(), kept.size() + spare.size());
  return kept;
}

}  // namespace

extern "C" {

void lgicp_default_config(lgicp_config* c, int base) {
  std::memset(c, 0, sizeof(*c));
  c->abi_version = LGICP_ABI_VERSION;
  c->device = 0;
  // loop_closure/config/laser_parameters.yaml, robot block :5-147, base block :153-301
  c->tf_epsilon = 1e-10;
  c->corr_dist = base ? 0.5 : 0.1;
  c->iterations = base ? 200 : 5;
  c->optimizer_iterations = 50;
  c->rotation_epsilon = 2e-3;
  c->k_correspondences = 20;
  c->gicp_epsilon = 1e-3;
  c->covariance_from_knn = 0;
  c->init_method = LGICP_INIT_FEATURES;
  c->covariance_method = LGICP_COV_POINT2PLANE;
  c->max_tolerable_fitness = base ? 0.9 : 0.4;
  c->transform_thresholding = base ? 1 : 0;
  c->max_translation = 20;
  c->max_rotation_deg = 50;
  c->num_prev_scans = 2;
  c->num_next_scans = 2;
  c->accumulate_source = base ? 0 : 1;
  c->laser_lc_rot_sigma = 0.01;
  c->laser_lc_trans_sigma = 0.01;
  c->use_fixed_covariances = 0;
  c->sac_iterations = base ? 500 : 10;
  c->sac_features_radius = 2.0;
  c->sac_fitness_score_threshold = base ? 32.0 : 1.2;
  c->harris_threshold = 0.000000005;
  c->harris_suppression = 1;
  c->harris_radius = 0.1;
  c->harris_refine = 0;
  c->harris_response = 1;
  c->workspace_bytes = 0;
}

int lgicp_create(const lgicp_config* cfg, lgicp_ctx** out) {
  if (!cfg || !out) return fail(nullptr, LGICP_E_INVALID, "null argument");
  if (cfg->abi_version != LGICP_ABI_VERSION)
    return fail(nullptr, LGICP_E_INVALID, "abi_version %d != %d", cfg->abi_version,
                LGICP_ABI_VERSION);
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(nullptr, LGICP_E_NO_DEVICE, "no HIP device visible");
  if (cfg->device < 0 || cfg->device >= n_dev)
    return fail(nullptr, LGICP_E_INVALID, "device %d out of range (%d devices)", cfg->device,
                n_dev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess)
    return fail(nullptr, LGICP_E_HIP, "hipGetDeviceProperties failed");
  if (cfg->solver != LGICP_SOLVER_GAUSS_NEWTON && cfg->solver != LGICP_SOLVER_BFGS)
    return fail(nullptr, LGICP_E_INVALID, "solver %d is not one of LGICP_SOLVER_*", cfg->solver);
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, LGICP_E_NO_DEVICE, "device %d is %s, this library is built for gfx950",
                cfg->device, prop.gcnArchName);
  if (hipSetDevice(cfg->device) != hipSuccess)
    return fail(nullptr, LGICP_E_HIP, "hipSetDevice failed");
  {
    const int rc = static_scratch_init(cfg->device);
    if (rc != LGICP_OK) return rc;
  }
  lgicp_ctx* ctx = new lgicp_ctx();
  ctx->cfg = *cfg;
  std::memset(&ctx->stats, 0, sizeof(ctx->stats));
  if (ctx->cfg.workspace_bytes == 0) ctx->cfg.workspace_bytes = (size_t)16 << 30;
  if (cfg->reserved[0] == 2 || cfg->reserved[0] == 4 || cfg->reserved[0] == 8)
    ctx->nn_variant = cfg->reserved[0];
  ctx->nn_exhaustive = cfg->reserved[2] == 1;
  if (cfg->reserved[3] > 0) {
    const float r0 = 1e-3f * (float)cfg->reserved[3];
    ctx->nn_r0sq = r0 * r0;
  }
  if (cfg->reserved[7] >= 1 && cfg->reserved[7] <= lgicp_ctx::MAX_SLOTS)
    ctx->n_slots = cfg->reserved[7];
  std::vector<hipStream_t> streams = concurrent_streams(ctx->n_slots, /*probe=*/cfg->reserved[7] < 1);
  ctx->n_slots = std::max<int>(1, std::min<int>(ctx->n_slots, (int)streams.size()));
  ctx->n_wide = ctx->n_slots;
  bool slots_ok = !streams.empty();
  for (int k = 0; k < ctx->n_slots && slots_ok; ++k) slots_ok = ctx->slot[k].create(streams[k]);
  if (!slots_ok) {
    for (size_t k = 0; k < streams.size(); ++k)  // the ones no slot has taken over
      if ((int)k >= ctx->n_slots || !ctx->slot[k].stream) (void)hipStreamDestroy(streams[k]);
    for (int k = 0; k < lgicp_ctx::MAX_SLOTS; ++k) ctx->slot[k].destroy();
    delete ctx;
    return fail(nullptr, LGICP_E_HIP, "stream / pinned allocation failed");
  }
  ctx->stream = ctx->slot[0].stream;
  ctx->feature_scratch = std::shared_ptr<FeatureScratch>(new FeatureScratch[lgicp_ctx::FEATURE_SETS * FEATURE_LANES],
                                                         std::default_delete<FeatureScratch[]>());
  ctx->sac_lanes = std::shared_ptr<SacLane>(new SacLane[lgicp_ctx::FEATURE_SETS * FEATURE_LANES],
                                            std::default_delete<SacLane[]>());
  ctx->helpers[0].start(ctx->n_slots - 1, ctx);
  *out = ctx;
  return LGICP_OK;
}

void lgicp_destroy(lgicp_ctx* ctx) {
  free_parked_device_memory(nullptr);
  if (!ctx) return;
  for (int l = 0; l < lgicp_ctx::MAX_LANES; ++l) ctx->helpers[l].shutdown();
  (void)hipSetDevice(ctx->cfg.device);
  ParkScope park(ctx);  // whatever is released below through the parking list stays with ctx
  free_parked_device_memory(ctx);
  for (CliqueBuffers*& b : ctx->clique_bufs) {
    clique_buffers_delete(b);
    b = nullptr;
  }
  for (int k = 0; k < lgicp_ctx::MAX_SLOTS; ++k) ctx->slot[k].destroy();
  ctx->acc_cache.clear();  // parked by their deleters, freed below with everything else parked
  for (auto& kv : ctx->scans) kv.second.free_device();
  ctx->reg.destroy();
  for (int f = 0; f < lgicp_ctx::FEATURE_SETS; ++f) {
    ctx->teaser_slots[f].clear();  // device buffers: released while the context's device is current
    if (ctx->matcher[f]) ctx->matcher[f]->release();
  }
  if (ctx->feature_scratch)
    for (int l = 0; l < lgicp_ctx::FEATURE_SETS * FEATURE_LANES; ++l) ctx->feature_scratch.get()[l].release();
  if (ctx->sac_lanes)
    for (int l = 0; l < lgicp_ctx::FEATURE_SETS * FEATURE_LANES; ++l) ctx->sac_lanes.get()[l].release();
  free_parked_device_memory(ctx);
  delete ctx;
}

const char* lgicp_last_error(const lgicp_ctx* ctx) {
  if (!ctx) return g_create_error.c_str();
  // a copy per calling thread: submits of other threads may set a new message meanwhile
  static thread_local std::string copy;
  std::lock_guard<std::mutex> lock(const_cast<lgicp_ctx*>(ctx)->err_mu);
  copy = ctx->err;
  return copy.c_str();
}

}  // extern "C"

namespace {

// Holds every lane of the context: nothing of it is in flight while this exists, and no submit
// starts.  Outgrown device buffers parked by earlier submits are freed on the way in.
struct AllLanes {
  lgicp_ctx* ctx;
  ParkScope park;
  LaneTable::Grant grant;
  explicit AllLanes(lgicp_ctx* c)
      : ctx(c), park(c), grant(c->lanes.acquire_all([c] { free_parked_device_memory(c); })) {}
  ~AllLanes() { ctx->lanes.release(grant); }
};

// One lane of the context -- or, when `wide` lanes from lane 0 on are wanted and no submit is in
// flight at all, those: a caller that is alone on a context with several lanes gets the width a
// single submit can use, callers that overlap get one lane each.  (Outgrown buffers are freed only
// when nothing is in flight: they may still be read by what another lane is running.)
struct OneLane {
  lgicp_ctx* ctx;
  ParkScope park;
  LaneTable::Grant grant;
  int lane, held;
  OneLane(lgicp_ctx* c, int wide)
      : ctx(c), park(c), grant(c->lanes.acquire_one(wide, [c] { free_parked_device_memory(c); })),
        lane(grant.lane), held(grant.held) {}
  ~OneLane() { ctx->lanes.release(grant); }
};

// caller holds data_mu exclusively
int add_keyframes_locked(lgicp_ctx* ctx, size_t count, const uint64_t* keys, const float* const* xyz,
                         const float* const* normals, const size_t* n_points, size_t stride_bytes) {
  if (count && (!keys || !xyz || !n_points)) return fail(ctx, LGICP_E_INVALID, "null argument");
  if (stride_bytes < 12 || stride_bytes % 4) return fail(ctx, LGICP_E_INVALID, "bad stride");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  // first insert of a key wins (IcpLoopComputation.cc:380-385), inside the call as well
  std::vector<KfInput> in;
  std::unordered_map<uint64_t, int> seen;
  for (size_t k = 0; k < count; ++k) {
    if (n_points[k] > 0 && !xyz[k]) return fail(ctx, LGICP_E_INVALID, "xyz is null");
    if (n_points[k] > (size_t)1 << 28) return fail(ctx, LGICP_E_INVALID, "cloud too large");
    if (ctx->scans.count(keys[k]) || seen.count(keys[k])) continue;
    seen[keys[k]] = 1;
    in.push_back({keys[k], xyz[k], normals ? normals[k] : nullptr, n_points[k], stride_bytes / 4});
  }
  // sub-batches of at most 4M points: 128 MiB of pinned staging, 100 MiB of sort scratch
  size_t first = 0;
  std::vector<Keyframe> kfs;
  while (first < in.size()) {
    size_t last = first, pts = 0;
    while (last < in.size() && (last == first || pts + in[last].n <= ((size_t)4 << 20))) pts += in[last++].n;
    kfs.assign(last - first, Keyframe());
    const int rc = register_keyframes(ctx, in.data() + first, last - first, kfs.data(), /*derive=*/true);
    if (rc != LGICP_OK) {
      // work of this sub-batch may be queued on buffers about to be freed
      (void)hipStreamSynchronize(ctx->stream);
      for (Keyframe& kf : kfs) kf.free_device();
      ctx->reg.pending.erase(std::remove_if(ctx->reg.pending.begin(), ctx->reg.pending.end(),
                                            [&](const Registry::Pending& p) {
                                              return !ctx->scans.count(p.key);
                                            }),
                             ctx->reg.pending.end());
      return rc;
    }
    for (size_t k = first; k < last; ++k) ctx->scans.emplace(in[k].key, kfs[k - first]);
    first = last;
  }
  return (int)in.size();
}

}  // namespace

extern "C" {

int lgicp_add_keyframes(lgicp_ctx* ctx, size_t count, const uint64_t* keys, const float* const* xyz,
                        const float* const* normals, const size_t* n_points, size_t stride_bytes) {
  if (!ctx) return LGICP_E_INVALID;
  std::lock_guard<std::mutex> gate(ctx->gate_mu);
  std::unique_lock<std::shared_mutex> data(ctx->data_mu);  // waits for submits in flight
  return add_keyframes_locked(ctx, count, keys, xyz, normals, n_points, stride_bytes);
}

int lgicp_set_submit_lanes(lgicp_ctx* ctx, int lanes) {
  if (!ctx) return LGICP_E_INVALID;
  if (lanes < 1) return fail(ctx, LGICP_E_INVALID, "lanes must be >= 1");
  std::lock_guard<std::mutex> gate(ctx->gate_mu);
  std::unique_lock<std::shared_mutex> data(ctx->data_mu);
  AllLanes all(ctx);
  lanes = std::min<int>(lanes, lgicp_ctx::MAX_LANES);
  const bool feature_init = ctx->cfg.init_method == LGICP_INIT_FEATURES ||
                            ctx->cfg.init_method == LGICP_INIT_TEASERPP;
  // a feature-based initialisation works with one of the context's sets of feature buffers and the
  // four streams that go with it
  if (feature_init) lanes = std::min<int>(lanes, lgicp_ctx::FEATURE_SETS);
  const int want_slots = feature_init && lanes > 1 ? FEATURE_LANES * lanes : lanes;
  // More lanes than the context has slots: further slots on streams of their own.  They share the
  // process's hardware queues with the first ones, which is what is wanted here (a lane's work is
  // a chain of dependent launches with host round trips in between; a queue serves another lane
  // meanwhile); one submit is never spread over them.
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  while (ctx->n_slots < want_slots) {
    hipStream_t s = nullptr;
    HIP_TRY(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (!ctx->slot[ctx->n_slots].create(s)) {
      ctx->slot[ctx->n_slots].destroy();
      return fail(ctx, LGICP_E_HIP, "pinned allocation failed");
    }
    ++ctx->n_slots;
  }
  for (int l = 0; l < lgicp_ctx::MAX_LANES; ++l) ctx->helpers[l].shutdown();
  ctx->lanes.set_lanes(lanes);
  // (lane 0's helpers also serve a lone submit that is given several lanes)
  for (int l = 0; l < lanes; ++l)
    ctx->helpers[l].start(feature_init ? 0 : std::max(ctx->n_slots / lanes, l == 0 ? ctx->n_wide : 1) - 1, ctx);
  return lanes;
}

int lgicp_add_keyframe(lgicp_ctx* ctx, uint64_t key, const float* xyz, const float* normals,
                       size_t n, size_t stride_bytes) {
  return lgicp_add_keyframes(ctx, 1, &key, &xyz, normals ? &normals : nullptr, &n, stride_bytes);
}

int lgicp_wait_registered(lgicp_ctx* ctx) {
  if (!ctx) return LGICP_E_INVALID;
  std::lock_guard<std::mutex> gate(ctx->gate_mu);
  std::unique_lock<std::shared_mutex> data(ctx->data_mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  const int rc = flush_registrations(ctx);
  if (rc != LGICP_OK) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return LGICP_OK;
}

int lgicp_add_pose(lgicp_ctx* ctx, uint64_t key, const double q[4], const double t[3]) {
  if (!ctx || !q || !t) return LGICP_E_INVALID;
  std::lock_guard<std::mutex> gate(ctx->gate_mu);
  std::unique_lock<std::shared_mutex> data(ctx->data_mu);
  if (ctx->poses.count(key)) return 0;  // first wins (IcpLoopComputation.cc:400-402)
  ctx->poses.emplace(key, hm::pose_from_quat(q, t));
  return 1;
}

int lgicp_has_keyframe(const lgicp_ctx* ctx, uint64_t key) {
  if (!ctx) return 0;
  std::shared_lock<std::shared_mutex> data(ctx->data_mu);
  return ctx->scans.count(key) ? 1 : 0;
}
int lgicp_has_pose(const lgicp_ctx* ctx, uint64_t key) {
  if (!ctx) return 0;
  std::shared_lock<std::shared_mutex> data(ctx->data_mu);
  return ctx->poses.count(key) ? 1 : 0;
}

int lgicp_get_stats(const lgicp_ctx* ctx, lgicp_stats* out) {
  if (!ctx || !out) return LGICP_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->stats_mu);
  *out = ctx->stats;
  return LGICP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
namespace {

struct PairPlan {
  size_t cand;  // index into the caller's arrays
  int src_cloud, tgt_cloud, scan1_cloud, scan2_cloud;
  float guess[16];
  hm::Pose pose21_odom;
};

struct BatchPlan {
  std::vector<CloudPlan> clouds;
  std::map<std::pair<uint64_t, int>, int> cloud_index;
  std::vector<PairPlan> pairs;
  size_t bytes = 0;
};

int plan_cloud(lgicp_ctx* ctx, BatchPlan& bp, uint64_t key, bool accumulate) {
  // AccumulateScans (IcpLoopComputation.cc:910-947): prev keys key-1.. then next keys key+1..;
  // a neighbour is used only when both its scan and its pose exist
  std::vector<std::pair<uint64_t, hm::Pose>> segs;
  if (accumulate) {
    const hm::Pose& new_pose = ctx->poses.at(key);
    for (int i = 0; i < ctx->cfg.num_prev_scans; ++i) {
      const uint64_t k = key - (uint64_t)i - 1;
      if (!ctx->poses.count(k) || !ctx->scans.count(k)) continue;
      segs.emplace_back(k, hm::between(new_pose, ctx->poses.at(k)));
    }
    for (int i = 0; i < ctx->cfg.num_next_scans; ++i) {
      const uint64_t k = key + (uint64_t)i + 1;
      if (!ctx->poses.count(k) || !ctx->scans.count(k)) continue;
      segs.emplace_back(k, hm::between(new_pose, ctx->poses.at(k)));
    }
  }
  const bool acc = !segs.empty();
  auto id = std::make_pair(key, acc ? 1 : 0);
  auto it = bp.cloud_index.find(id);
  if (it != bp.cloud_index.end()) return it->second;
  CloudPlan cp;
  cp.key = key;
  cp.accumulated = acc;
  cp.segs = std::move(segs);
  cp.n = ctx->scans.at(key).n;
  for (auto& s : cp.segs) cp.n += ctx->scans.at(s.first).n;
  cp.n_pad = std::max(NN_TILE, round_up(cp.n, NN_TILE));
  if (acc && !ctx->cfg.covariance_from_knn && ctx->cfg.accumulated_cache_bytes != 1) {
    // every member finite (the k-d order of the whole needs the count of finite points), and no
    // normals to be re-estimated over the accumulated cloud for the feature front end
    const Keyframe& own = ctx->scans.at(key);
    const bool feature_init = ctx->cfg.init_method == LGICP_INIT_FEATURES ||
                              ctx->cfg.init_method == LGICP_INIT_TEASERPP;
    bool ok = own.nf == own.n && !(own.normals_missing && feature_init);
    for (auto& sg : cp.segs) ok = ok && ctx->scans.at(sg.first).nf == ctx->scans.at(sg.first).n;
    cp.cacheable = ok;
  }
  if ((acc && !cp.cacheable) || ctx->cfg.covariance_from_knn) bp.bytes += (size_t)cp.n * (48 + 64) + 1024;  // covariances, records (or flagged normals + compact records)
  if (acc && !cp.cacheable) {
    bp.bytes += (size_t)cp.n_pad * 16 + (size_t)cp.n * 32 + 1024;  // xyz + normals (+ estimated)
    // spatial index: tile-ordered points, boxes, visiting order
    bp.bytes += (size_t)cp.n * 23 + (cp.segs.size() + 1) * (SI_TILE * 16 + 64) + 4096;
  }
  const int idx = (int)bp.clouds.size();
  bp.clouds.push_back(std::move(cp));
  bp.cloud_index[id] = idx;
  return idx;
}

void guess_from_pose(const hm::Pose& p, bool with_translation, float g[16]) {
  for (int i = 0; i < 16; ++i) g[i] = (i % 5 == 0) ? 1.f : 0.f;
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) g[a * 4 + b] = (float)p.R[a * 3 + b];
    if (with_translation) g[a * 4 + 3] = (float)p.t[a];
  }
}

// runs one device batch; results for bp.pairs[i] go to out[bp.pairs[i].cand]
// Device-side description of every distinct cloud of a batch.  Keyframes that need no
// accumulation point straight at their resident buffers; an accumulated cloud gets arena space
// for points, normals, covariances and an index that is the concatenation of its members'
// segments, plus the copy/transform segments (AccumulateScans, IcpLoopComputation.cc:910-947)
// that will fill it.
struct CloudTables {
  std::vector<CloudDev> clouds;
  std::vector<SegmentDev> segs;
  std::vector<IndexSegDev> isegs;
  std::vector<int> normal_jobs;  // accumulated clouds whose normals have to be estimated
  int max_seg_n = 0, max_cloud_n = 0, max_iseg_n = 0;
};

void build_cloud_tables(lgicp_ctx* ctx, BatchPlan& bp, Arena& ar, CloudTables* t) {
  const lgicp_config& cfg = ctx->cfg;
  const int n_clouds = (int)bp.clouds.size();
  t->clouds.resize((size_t)n_clouds);
  std::vector<CloudDev>& h_clouds = t->clouds;
  std::vector<SegmentDev>& h_segs = t->segs;
  std::vector<IndexSegDev>& h_isegs = t->isegs;
  std::vector<int>& normal_jobs = t->normal_jobs;
  int& max_seg_n = t->max_seg_n;
  int& max_cloud_n = t->max_cloud_n;
  int& max_iseg_n = t->max_iseg_n;
  for (int c = 0; c < n_clouds; ++c) {
    CloudPlan& cp = bp.clouds[c];
    const Keyframe& kf = cp.entry ? cp.entry->kf : ctx->scans.at(cp.key);
    CloudDev& cd = h_clouds[c];
    std::memset(&cd, 0, sizeof(cd));
    cd.n = cp.n;
    cd.n_pad = cp.n_pad;
    const bool own_cov = cp.in_arena() || cfg.covariance_from_knn;
    cd.cov = own_cov ? ar.take<double>((size_t)cp.n * 6) : kf.d_cov;
    // covariances from normals: flagged normals and 32-byte records; from k-NN: 64-byte records
    cd.compact = cfg.covariance_from_knn ? 0 : 1;
    cd.rec = own_cov ? ar.take<float4>((size_t)cp.n * (cd.compact ? 2 : 4)) : kf.d_rec;
    cd.cnrm = cd.compact ? (own_cov ? ar.take<float4>((size_t)cp.n) : kf.d_cnrm) : nullptr;
    max_cloud_n = std::max(max_cloud_n, cp.n);
    if (!cp.in_arena()) {
      cd.xyz = kf.d_xyz;
      cd.nrm = kf.d_nrm;
      cd.nrm_x = kf.d_nrm_x;
      std::memcpy(cd.centroid, kf.centroid, sizeof(cd.centroid));
      cd.factor = kf.factor;
      cd.sxyz = kf.d_sxyz, cd.tmin = kf.d_tmin, cd.tmax = kf.d_tmax, cd.qorder = kf.d_qorder;
      cd.n_tiles = kf.n_tiles, cd.n_q = kf.n_q;
    } else {
      float4* xyz = ar.take<float4>((size_t)cp.n_pad);
      float4* nrm = ar.take<float4>((size_t)cp.n);
      cd.xyz = xyz;
      cd.nrm = nrm;
      cd.nrm_x = nrm;
      if (kf.normals_missing && (cfg.init_method == LGICP_INIT_FEATURES ||
                                 cfg.init_method == LGICP_INIT_TEASERPP)) {
        // the feature front end extracts normals from the accumulated cloud
        // (IcpLoopComputation.cc:654-658): recomputed over all of its points
        float4* nx = ar.take<float4>((size_t)cp.n);
        cd.nrm_x = nx;
        normal_jobs.push_back(c);
      }
      {  // index = concatenation of the member keyframes' segments
        int tiles = (kf.n + SI_TILE - 1) / SI_TILE;
        for (auto& sg : cp.segs) tiles += (ctx->scans.at(sg.first).n + SI_TILE - 1) / SI_TILE;
        cd.n_tiles = tiles;
        cd.n_q = round_up(cp.n, 64);
        float4* sx = ar.take<float4>((size_t)tiles * SI_TILE);
        float4* t0 = ar.take<float4>((size_t)tiles + (tiles + 63) / 64);  // tiles, blocks of 64 tiles
        float4* t1 = ar.take<float4>((size_t)tiles + (tiles + 63) / 64);
        int* qo = ar.take<int>((size_t)cd.n_q);
        cd.sxyz = sx, cd.tmin = t0, cd.tmax = t1, cd.qorder = qo;
        IndexSegDev g;
        std::memset(&g, 0, sizeof(g));
        g.xyz = xyz, g.sxyz = sx, g.tmin = t0, g.tmax = t1, g.qorder = qo, g.n_q = cd.n_q;
        int poff = 0, toff = 0;
        auto add = [&](const Keyframe& k, bool last) {
          g.order = k.d_order, g.point_off = poff, g.tile_off = toff, g.n = k.n, g.last = last;
          if (k.n > 0 || last) h_isegs.push_back(g);
          max_iseg_n = std::max(max_iseg_n, k.n);
          poff += k.n;
          toff += (k.n + SI_TILE - 1) / SI_TILE;
        };
        add(kf, cp.segs.empty());
        for (size_t si = 0; si < cp.segs.size(); ++si)
          add(ctx->scans.at(cp.segs[si].first), si + 1 == cp.segs.size());
      }
      int off = 0;
      SegmentDev s;
      std::memset(&s, 0, sizeof(s));
      double* const acc_cov = cfg.covariance_from_knn ? nullptr : const_cast<double*>(cd.cov);
      s.src_xyz = kf.d_xyz;
      s.src_nrm = kf.d_nrm;
      s.dst_xyz = xyz;
      s.dst_nrm = nrm;
      s.src_cov = acc_cov ? kf.d_cov : nullptr;
      s.dst_cov = acc_cov;
      s.src_cnrm = acc_cov ? kf.d_cnrm : nullptr;
      s.dst_cnrm = acc_cov ? const_cast<float4*>(cd.cnrm) : nullptr;
      s.dst_rec = acc_cov ? const_cast<float4*>(cd.rec) : nullptr;
      s.n = kf.n;
      s.identity = 1;
      h_segs.push_back(s);
      max_seg_n = std::max(max_seg_n, s.n);
      off += kf.n;
      for (auto& sg : cp.segs) {
        const Keyframe& nk = ctx->scans.at(sg.first);
        std::memset(&s, 0, sizeof(s));
        s.src_xyz = nk.d_xyz;
        s.src_nrm = nk.d_nrm;
        s.dst_xyz = xyz + off;
        s.dst_nrm = nrm + off;
        s.src_cov = acc_cov ? nk.d_cov : nullptr;
        s.dst_cov = acc_cov ? acc_cov + 6 * (size_t)off : nullptr;
        s.src_cnrm = acc_cov ? nk.d_cnrm : nullptr;
        s.dst_cnrm = acc_cov ? const_cast<float4*>(cd.cnrm) + off : nullptr;
        s.dst_rec = acc_cov ? const_cast<float4*>(cd.rec) + 2 * (size_t)off : nullptr;
        s.n = nk.n;
        s.identity = 0;
        for (int a = 0; a < 3; ++a) {
          for (int b = 0; b < 3; ++b) s.T[a * 4 + b] = sg.second.R[a * 3 + b];
          s.T[a * 4 + 3] = sg.second.t[a];
        }
        h_segs.push_back(s);
        max_seg_n = std::max(max_seg_n, s.n);
        off += nk.n;
      }
    }
  }

}

// FEATURES / TEASERPP initialisation (IcpLoopComputation.cc:517-531, 650-705, 949-1081): Harris
// keypoints and FPFH descriptors once per distinct cloud of the batch, then SAC-IA or TEASER++
// per pair starting from the odometry guess.  Rewrites the pairs' guesses (and the state of
// pairs that SAC-IA rejects: keypoint fitness at or above the threshold) in the host tables;
// the caller uploads them again.
int initialise_from_features(lgicp_ctx* ctx, const BatchPlan& bp, const std::vector<CloudDev>& h_clouds,
                             const lgicp_candidate* cand, lgicp_result* out,
                             std::vector<PairDev>& h_pairs, std::vector<PairState>& h_state,
                             std::vector<char>& init_rejected, lgicp_stats& stats, int fset) {
  const lgicp_config& cfg = ctx->cfg;
  const int n_pairs = (int)bp.pairs.size();
  if (fset < 0 || fset >= lgicp_ctx::FEATURE_SETS || ctx->n_slots < FEATURE_LANES * (fset + 1)) fset = 0;
  const int lane0 = FEATURE_LANES * fset;  // this set's streams: slots lane0 .. lane0 + 3
  hipStream_t main_stream = ctx->slot[lane0 < ctx->n_slots ? lane0 : 0].stream;
  // Keyframes keep their features: the look-up and the computation of the missing ones belong to
  // one submit at a time (a second one waits here, then finds them)
  std::unique_lock<std::mutex> feat_lock(ctx->feat_mu);
  std::map<int, std::shared_ptr<CloudFeatures>> feat_ptr;
  std::vector<int> todo;  // clouds whose features have to be computed
  for (int p = 0; p < n_pairs; ++p)
    for (int c : {bp.pairs[p].src_cloud, bp.pairs[p].tgt_cloud})
      if (!feat_ptr.count(c)) {
        // a keyframe on its own keeps its features for later submits; an accumulated cloud is
        // assembled per batch and so are its features
        // (an accumulated cloud kept on the device keeps its features with it)
        Keyframe& kf = bp.clouds[c].entry ? bp.clouds[c].entry->kf : ctx->scans.at(bp.clouds[c].key);
        std::shared_ptr<CloudFeatures> f = bp.clouds[c].in_arena() ? nullptr : kf.features;
        if (!f) {
          f = std::make_shared<CloudFeatures>();
          todo.push_back(c);
          if (!bp.clouds[c].in_arena()) kf.features = f;
        }
        feat_ptr[c] = f;
      }
  if (!todo.empty()) {
    // One cloud's sweeps (16k points: a few hundred waves) fill a fraction of the GPU, so several
    // clouds go side by side: one host thread, stream and scratch set per lane.
    const int lanes = std::max(1, std::min({(int)todo.size(), FEATURE_LANES, ctx->n_slots}));
    std::atomic<size_t> next{0};
    std::atomic<int> rc_all{LGICP_OK};
    auto run = [&](int lane) {
      if (hipSetDevice(cfg.device) != hipSuccess) rc_all.store(LGICP_E_HIP);
      for (;;) {
        const size_t q = next.fetch_add(1);
        if (q >= todo.size() || rc_all.load() != LGICP_OK) return;
        const int c = todo[q];
        const int rc = compute_cloud_features(ctx, h_clouds[c].xyz, h_clouds[c].nrm_x, h_clouds[c].n,
                                              feat_ptr[c].get(), lane0 + lane);
        if (rc != LGICP_OK) rc_all.store(rc);
      }
    };
    auto run_parked = [&](int lane) {  // on a thread of its own: buffers it outgrows stay with ctx
      ParkScope park(ctx);
      run(lane);
    };
    std::vector<std::thread> pool;
    for (int l = 1; l < lanes; ++l) pool.emplace_back(run_parked, l);
    run(0);
    for (std::thread& t : pool) t.join();
    if (rc_all.load() != LGICP_OK) {
      // nothing half-made stays behind on a keyframe
      for (int c : todo)
        if (!bp.clouds[c].in_arena())
          (bp.clouds[c].entry ? bp.clouds[c].entry->kf : ctx->scans.at(bp.clouds[c].key)).features.reset();
      {
        bool no_text;
        {
          std::lock_guard<std::mutex> lock(ctx->err_mu);  // other lanes may be setting a message
          no_text = ctx->err.empty();
        }
        if (rc_all.load() == LGICP_E_HIP && no_text) fail(ctx, LGICP_E_HIP, "feature front end failed");
      }
      return rc_all.load();
    }
  }
  feat_lock.unlock();
  struct FeatView {  // feats[c] as before
    std::map<int, std::shared_ptr<CloudFeatures>>& m;
    const CloudFeatures& operator[](int c) const { return *m.at(c); }
  } feats{feat_ptr};
  if (cfg.init_method == LGICP_INIT_TEASERPP) {
    // IcpLoopComputation.cc:528-531: never rejects; keeps the odometry guess unless the
    // TEASER++ transform explains the descriptor matches at least as well.  Pairs are taken in
    // groups: matches and consistency graph pair by pair, then clique search and rotation search
    // for the whole group together (their steps are latency-bound launches, which cost the same
    // for one pair as for sixty-four).  Largest pairs first, so that the pairs of a group are of
    // similar size and the buffers of the slots are sized once.
    const bool trace = std::getenv("LGICP_TRACE") != nullptr;  // diagnostics on stderr
    constexpr int kGroup = 512;
    // the first group of a submit has nothing to run beside (its matches and graphs are waited
    // for), so it is kept small: the pairs with the longest chains, whose search then hides the
    // preparation of all the others
    constexpr int kFirstGroup = 64;
    constexpr size_t kGroupBytes = 12ull << 30;  // bit matrices of a group
    // Two sets of slots: while the clique search of one group runs on the context's stream (chains
    // of launches that leave much of the GPU idle in their tails), a helper thread enqueues the
    // matches and graphs of the next group on a second stream.
    std::vector<std::shared_ptr<TeaserJob>>& slots = ctx->teaser_slots[fset];
    while ((int)slots.size() < 2 * kGroup) slots.push_back(std::make_shared<TeaserJob>());
    // streams the matches and graphs of a group are enqueued on, pair after pair in turn (a pair's
    // kernels are small: a few of them side by side fill the GPU better): the set's streams other
    // than the one the clique search runs on
    hipStream_t prep_streams[FEATURE_LANES - 1] = {main_stream, main_stream, main_stream};
    const int n_prep = std::max(1, std::min(FEATURE_LANES - 1, ctx->n_slots - lane0 - 1));
    for (int j = 0; j < n_prep && ctx->n_slots >= lane0 + 2; ++j) prep_streams[j] = ctx->slot[lane0 + 1 + j].stream;
    std::vector<int> by_size((size_t)n_pairs);
    for (int p = 0; p < n_pairs; ++p) by_size[p] = p;
    std::stable_sort(by_size.begin(), by_size.end(), [&](int a, int b) {
      return feats[bp.pairs[a].src_cloud].idx.size() > feats[bp.pairs[b].src_cloud].idx.size();
    });
    struct Group {
      int count = 0, rc = LGICP_OK;
      double ms = 0;
      int pair[kGroup];
      TeaserJob* jobs[kGroup];
      const float* guesses[kGroup];
    };
    int q = 0;  // next pair (in by_size order) to prepare; touched by one thread at a time
    auto prepare = [&](Group& G, int half, bool own_thread) {
      const auto t_a = std::chrono::steady_clock::now();
      G.count = 0, G.rc = LGICP_OK;
      if (own_thread && hipSetDevice(cfg.device) != hipSuccess) {
        G.rc = fail(ctx, LGICP_E_HIP, "hipSetDevice failed");
        return;
      }
      size_t bytes = 0;
      const int room = q == 0 ? kFirstGroup : kGroup;
      while (q < n_pairs && G.count < room && bytes < kGroupBytes) {
        const int p = by_size[q++];
        const PairPlan& pp = bp.pairs[p];
        TeaserJob* job = slots[(size_t)half * kGroup + G.count].get();
        const int sub = G.count % n_prep;
        G.rc = teaser_prepare_device(ctx, feats[pp.src_cloud], feats[pp.tgt_cloud], job, prep_streams[sub], fset, sub);
        if (G.rc != LGICP_OK) {  // nothing of this group stays in flight behind the error
          for (int j = 0; j < n_prep; ++j) (void)hipStreamSynchronize(prep_streams[j]);
          return;
        }
        bytes += (size_t)job->graph.n * job->graph.words * sizeof(uint64_t);
        G.pair[G.count] = p, G.jobs[G.count] = job, G.guesses[G.count] = pp.guess;
        ++G.count;
      }
      for (int j = 0; j < n_prep; ++j)
        if (hipStreamSynchronize(prep_streams[j]) != hipSuccess) {
          G.rc = fail(ctx, LGICP_E_HIP, "TEASER++ initialisation: matches and graphs failed on the device (%s)",
                      hipGetErrorString(hipGetLastError()));
          return;
        }
      for (int j = 0; j < G.count; ++j) teaser_prepare_collect(G.jobs[j]);
      G.ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_a).count();
    };
    std::unique_ptr<Group> cur(new Group), nxt(new Group);
    int half = 0;
    prepare(*cur, half, false);
    while (cur->rc == LGICP_OK && cur->count > 0) {
      const auto t_b = std::chrono::steady_clock::now();
      std::thread helper;
      nxt->count = 0, nxt->rc = LGICP_OK;
      if (q < n_pairs)
        helper = std::thread([&] {
          ParkScope park(ctx);
          prepare(*nxt, half ^ 1, true);
        });
      const int g = cur->count;
      float T0[kGroup][16];
      int used[kGroup], nc[kGroup];
      CliqueTimes ct;
      const int rc = teaser_finish_group(ctx, cur->jobs, cur->guesses, g, T0, used, nc, &ct, main_stream, fset);
      if (helper.joinable()) helper.join();
      stats.clique_ms += ct.starts_ms, stats.clique_row_bytes += ct.starts_row_bytes;
      stats.clique_launches += ct.starts_launches;
      stats.core_ms += ct.core_ms, stats.core_matrix_bytes += ct.core_matrix_bytes;
      stats.core_launches += ct.core_launches;
      for (int j = 0; j < g; ++j)
        stats.teaser_oversized_pairs += cur->jobs[j]->n_matches > LGICP_TEASER_MAX_MATCHES;
      if (rc != LGICP_OK) return rc;
      for (int j = 0; j < g; ++j) {
        const int p = cur->pair[j];
        out[bp.pairs[p].cand].init_fitness = used[j] ? (double)nc[j] : 0.0;
        std::memcpy(h_pairs[p].guess, T0[j], sizeof(T0[j]));
        if (!h_state[p].active) std::memcpy(h_state[p].Tfinal, T0[j], sizeof(T0[j]));
      }
      if (trace)
        std::fprintf(stderr, "[lgicp] teaser group of %d pairs (%d-%d matches): graphs %.1f ms (beside the group "
                     "before), solve %.1f ms\n", g, cur->jobs[g - 1]->n_matches, cur->jobs[0]->n_matches, cur->ms,
                     1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_b).count());
      std::swap(cur, nxt);
      half ^= 1;
    }
    if (cur->rc != LGICP_OK) return cur->rc;
    // the slots keep their bit matrices for the next submit unless they have grown large
    size_t held = 0;
    for (auto& sl : slots) held += sl->d_bits.cap * sizeof(unsigned long long);
    if (held > (32ull << 30))
      for (auto& sl : slots) sl->d_bits.release();
  }
  if (cfg.init_method == LGICP_INIT_FEATURES && n_pairs > 0) {
    // A pair's alignment is two small launches with a host step between them (hypotheses from the
    // 5-NN table): several pairs go side by side, one host thread, stream and buffer set per lane.
    const int lanes = std::max(1, std::min({n_pairs, FEATURE_LANES, ctx->n_slots}));
    std::atomic<int> next{0}, rc_all{LGICP_OK};
    auto run = [&](int lane) {
      if (hipSetDevice(cfg.device) != hipSuccess) rc_all.store(LGICP_E_HIP);
      for (;;) {
        const int p = next.fetch_add(1);
        if (p >= n_pairs || rc_all.load() != LGICP_OK) return;
        const PairPlan& pp = bp.pairs[p];
        float T0[16];
        double score = 0;
        // one fixed sampling sequence per (key_from, key_to): results do not depend on how the
        // candidates were batched
        const uint32_t seed = 0x5AC1A000u ^ (uint32_t)(cand[pp.cand].key_from * 2654435761ull) ^
                              (uint32_t)(cand[pp.cand].key_to * 40503ull);
        const int rc = sac_align_device(ctx, feats[pp.src_cloud], feats[pp.tgt_cloud], pp.guess, seed, T0,
                                        &score, lane0 + lane);
        if (rc != LGICP_OK) {
          rc_all.store(rc);
          return;
        }
        out[pp.cand].init_fitness = score;
        std::memcpy(h_pairs[p].guess, T0, sizeof(T0));
        if (!(score < cfg.sac_fitness_score_threshold)) {
          init_rejected[p] = 1;
          h_state[p].active = 0;
          h_state[p].failed = 1;
        }
        if (!h_state[p].active) std::memcpy(h_state[p].Tfinal, T0, sizeof(T0));
      }
    };
    auto run_parked = [&](int lane) {
      ParkScope park(ctx);
      run(lane);
    };
    std::vector<std::thread> pool;
    for (int l = 1; l < lanes; ++l) pool.emplace_back(run_parked, l);
    run(0);
    for (std::thread& t : pool) t.join();
    if (rc_all.load() != LGICP_OK) return rc_all.load();
  }
  return LGICP_OK;
}

// What PerformAlignment does with a finished alignment (IcpLoopComputation.cc:568-640): status
// gates (finite, converged, fitness, transform thresholds), the inverse pose and its covariance
// / information matrix.  O(1) 6x6 algebra per pair, on the host as in the reference.  h_post
// holds the fitness / Ap partial sums of the batch, [pair][RED_MAX_CHUNKS][NPOST].
void finish_pairs(const lgicp_config& cfg, const BatchPlan& bp, const std::vector<PairDev>& h_pairs,
                  const std::vector<PairState>& h_state, const std::vector<int>& h_ints,
                  const double* h_post, const std::vector<char>& init_rejected, lgicp_result* out,
                  lgicp_stats& S) {
  const int n_pairs = (int)bp.pairs.size();
  for (int p = 0; p < n_pairs; ++p) {
    const PairPlan& pp = bp.pairs[p];
    const PairState& s = h_state[p];
    lgicp_result& r = out[pp.cand];
    r.iterations = s.iterations;
    r.converged = s.converged;
    r.n_correspondences = s.m;
    std::memcpy(r.T_icp, s.Tfinal, sizeof(r.T_icp));
    std::memcpy(r.T_init, h_pairs[p].guess, sizeof(r.T_init));
    if (init_rejected[p]) {
      r.status = LGICP_PAIR_INIT_REJECTED;
      continue;
    }
    S.outer_iterations += (uint64_t)s.iterations;
    S.reduce_points += (double)h_pairs[p].n_src * (double)s.iterations;
    S.reduce_inliers += s.m_sum;
    S.cycle_exits += (uint64_t)(s.cycle != 0);
    S.nn_distance_evals +=
        (double)h_pairs[p].n_src * (double)h_pairs[p].n_tgt * (double)(s.iterations + 1);
    double post[NPOST] = {0};
    for (int c = 0; c < h_ints[p]; ++c)
      for (int k = 0; k < NPOST; ++k) post[k] += h_post[((size_t)p * RED_MAX_CHUNKS + c) * NPOST + k];
    const float* T = s.Tfinal;
    bool finite = post[2] == 0.0;
    for (int i = 0; i < 16; ++i) finite = finite && std::isfinite(T[i]);
    if (cfg.covariance_method == LGICP_COV_POINT2PLANE && !finite) {
      r.status = LGICP_PAIR_NONFINITE;
      continue;
    }
    hm::Pose delta;
    for (int a = 0; a < 3; ++a) {
      for (int b = 0; b < 3; ++b) delta.R[a * 3 + b] = (double)T[a * 4 + b];
      delta.t[a] = (double)T[a * 4 + 3];
    }
    r.fitness = post[1] > 0 ? post[0] / post[1] : 1.7976931348623157e308;
    if (!s.converged) {
      r.status = LGICP_PAIR_NOT_CONVERGED;
      continue;
    }
    if (r.fitness > cfg.max_tolerable_fitness) {
      r.status = LGICP_PAIR_FITNESS;
      continue;
    }
    {
      double xi[6];
      hm::logmap(hm::between(pp.pose21_odom, delta), xi);
      const double trans_diff = std::sqrt(xi[3] * xi[3] + xi[4] * xi[4] + xi[5] * xi[5]);
      const double rot_diff = std::sqrt(xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2]);
      if (cfg.transform_thresholding &&
          (trans_diff > cfg.max_translation || rot_diff > cfg.max_rotation_deg * M_PI / 180.0)) {
        r.status = LGICP_PAIR_TRANSFORM;
        continue;
      }
    }
    const hm::Pose inv = hm::inverse(delta);
    std::memcpy(r.delta_R, inv.R, sizeof(r.delta_R));
    std::memcpy(r.delta_t, inv.t, sizeof(r.delta_t));
    hm::quat_from_R(inv.R, r.delta_q);
    double Ap[36];
    {
      int k = 3;
      for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) Ap[a * 6 + b] = Ap[b * 6 + a] = post[k++];
    }
    if (cfg.use_fixed_covariances) {
      hm::fixed_covariance(cfg.laser_lc_rot_sigma, cfg.laser_lc_trans_sigma, r.covariance);
      hm::inverse6(r.covariance, r.information);
    } else if (cfg.covariance_method == LGICP_COV_POINT2POINT) {
      hm::point_point_covariance(Ap /* = H */, r.fitness, r.covariance);
      hm::inverse6(r.covariance, r.information);
    } else {
      hm::point_plane_covariance(Ap, cfg.laser_lc_rot_sigma, cfg.laser_lc_trans_sigma,
                                 r.covariance);
      for (int i = 0; i < 36; ++i) r.information[i] = Ap[i] / (0.05 * 0.05);
    }
    r.status = LGICP_PAIR_OK;
    S.pairs_ok++;
  }
}

// Waits for an event by polling.  How hipEventSynchronize waits is a property of the device
// flags the process was started with: a host application that asked for blocking waits (an
// interrupt and a wake-up per wait, tens of microseconds) would add that to every outer
// iteration of every part, with the GPU idle meanwhile.
hipError_t spin_until(hipEvent_t ev) {
  for (;;) {
    const hipError_t e = hipEventQuery(ev);
    if (e != hipErrorNotReady) return e;
    __builtin_ia32_pause();
  }
}

// Accumulated clouds kept on the device (AccEntry, lgicp_config::accumulated_cache_bytes): every
// cacheable cloud of the submit's plans gets its entry -- found in the context's cache, or built
// now, once per key, on the stream and with the registration buffers of slot W, and waited for
// (the cold path: a cloud is built the first time a candidate names it, or again when a neighbour
// scan has arrived since).  Least recently used entries go when the budget is exceeded; a batch
// that still works with one keeps it alive until its slot's next batch, and the memory is freed
// through the parking list.
int resolve_accumulated(lgicp_ctx* ctx, std::vector<BatchPlan>& plans, BatchSlot* W) {
  std::map<uint64_t, std::shared_ptr<AccEntry>> fresh;
  std::vector<AccBuild> builds;
  {
    std::lock_guard<std::mutex> lk(ctx->acc_mu);
    for (BatchPlan& bp : plans)
      for (CloudPlan& cp : bp.clouds) {
        if (!cp.cacheable) continue;
        std::vector<uint64_t> members;
        for (auto& sg : cp.segs) members.push_back(sg.first);
        auto it = ctx->acc_cache.find(cp.key);
        if (it != ctx->acc_cache.end() && it->second->members == members) {
          cp.entry = it->second;
          cp.entry->last_use = ++ctx->acc_tick;
          continue;
        }
        auto f = fresh.find(cp.key);
        if (f != fresh.end()) {  // another part of this submit names the same cloud
          cp.entry = f->second;
          continue;
        }
        cp.entry = std::make_shared<AccEntry>();
        cp.entry->members = std::move(members);
        fresh[cp.key] = cp.entry;
        AccBuild b;
        b.own = &ctx->scans.at(cp.key);
        for (auto& sg : cp.segs) b.segs.emplace_back(&ctx->scans.at(sg.first), sg.second);
        b.out = cp.entry.get();
        builds.push_back(std::move(b));
      }
  }
  if (builds.empty()) return LGICP_OK;
  {
    const int rc = register_accumulated(ctx, W->acc_reg, W->stream, builds.data(), builds.size());
    if (rc != LGICP_OK) return rc;
  }
  HIP_TRY(ctx, hipStreamSynchronize(W->stream));
  std::lock_guard<std::mutex> lk(ctx->acc_mu);
  for (auto& f : fresh) {
    auto it = ctx->acc_cache.find(f.first);
    if (it != ctx->acc_cache.end()) ctx->acc_bytes -= it->second->bytes;
    f.second->last_use = ++ctx->acc_tick;
    ctx->acc_cache[f.first] = f.second;
    ctx->acc_bytes += f.second->bytes;
  }
  const size_t budget = ctx->cfg.accumulated_cache_bytes ? ctx->cfg.accumulated_cache_bytes : ((size_t)32 << 30);
  while (ctx->acc_bytes > budget && ctx->acc_cache.size() > 1) {
    auto lru = ctx->acc_cache.begin();
    for (auto it = ctx->acc_cache.begin(); it != ctx->acc_cache.end(); ++it)
      if (it->second->last_use < lru->second->last_use) lru = it;
    ctx->acc_bytes -= lru->second->bytes;
    ctx->acc_cache.erase(lru);
  }
  return LGICP_OK;
}

int run_batch_impl(lgicp_ctx* ctx, BatchSlot* W, BatchPlan& bp, const lgicp_candidate* cand,
                   lgicp_result* out, bool explicit_guess) {
  const bool trace_t = std::getenv("LGICP_TRACE") != nullptr;
  const auto t_enter = std::chrono::steady_clock::now();
  auto since = [&](const char* what) {
    if (trace_t)
      std::fprintf(stderr, "[lgicp] slot %d %s at %.3f ms\n", (int)(W - ctx->slot), what,
                   1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enter).count());
  };

  const lgicp_config& cfg = ctx->cfg;
  const bool stage_events = cfg.stage_timing != 0 || trace_t;  // events between the kernels
  hipStream_t st = W->stream;
  const int n_pairs = (int)bp.pairs.size();
  const int n_clouds = (int)bp.clouds.size();
  if (n_pairs == 0) return LGICP_OK;

  // ---- device scratch ----------------------------------------------------------------------
  size_t pair_bytes = 0;
  for (auto& p : bp.pairs) pair_bytes += pair_scratch_bytes(ctx->cfg, bp.clouds[p.src_cloud].n);
  HIP_TRY(ctx, W->arena.reserve(bp.bytes + pair_bytes + 4096));
  Arena ar{W};

  // accumulated clouds kept on the device (resolve_accumulated): alive at least until this slot's
  // next batch (the slot's previous batch has been waited for: what it kept alive may go)
  W->acc_in_use.clear();
  for (CloudPlan& cp : bp.clouds)
    if (cp.entry) W->acc_in_use.push_back(cp.entry);

  CloudTables tables;
  build_cloud_tables(ctx, bp, ar, &tables);
  std::vector<CloudDev>& h_clouds = tables.clouds;
  std::vector<SegmentDev>& h_segs = tables.segs;
  std::vector<IndexSegDev>& h_isegs = tables.isegs;
  const std::vector<int>& normal_jobs = tables.normal_jobs;
  const int max_seg_n = tables.max_seg_n, max_iseg_n = tables.max_iseg_n;

  // the large host tables live in the slot: their pages stay mapped from batch to batch
  std::vector<PairDev>& h_pairs = W->v_pairs;
  std::vector<PairState>& h_state = W->v_state;
  h_pairs.resize((size_t)n_pairs);
  h_state.resize((size_t)n_pairs);
  int max_src_n = 0;
  for (int p = 0; p < n_pairs; ++p) {
    const PairPlan& pp = bp.pairs[p];
    PairDev& pd = h_pairs[p];
    std::memset(&pd, 0, sizeof(pd));
    pd.src_cloud = pp.src_cloud;
    pd.tgt_cloud = pp.tgt_cloud;
    pd.scan1_cloud = pp.scan1_cloud;
    pd.scan2_cloud = pp.scan2_cloud;
    pd.n_src = bp.clouds[pp.src_cloud].n;
    pd.n_tgt = bp.clouds[pp.tgt_cloud].n;
    pd.out = ar.take<float4>((size_t)pd.n_src);
    pd.keys = ar.take<nnkey_t>((size_t)pd.n_src);
    // reserved[6] = 2: seeds but no certificates (diagnostics)
    pd.cert = (ctx->nn_exhaustive || cfg.reserved[6] == 2) ? nullptr : ar.take<float4>((size_t)pd.n_src);
    if (cfg.solver == LGICP_SOLVER_BFGS) {
      pd.bf_a = ar.take<float4>((size_t)pd.n_src);
      pd.bf_b = ar.take<float2>((size_t)pd.n_src);
      pd.bf_m = ar.take<double2>((size_t)pd.n_src * 3);
    }
    std::memcpy(pd.guess, pp.guess, sizeof(pd.guess));
    max_src_n = std::max(max_src_n, pd.n_src);
    PairState& s = h_state[p];
    std::memset(&s, 0, sizeof(s));
    for (int i = 0; i < 16; ++i) s.T[i] = s.Tprev[i] = s.Tfinal[i] = (i % 5 == 0) ? 1.f : 0.f;
    s.active = 1;
    // computeCovariances returns early when a cloud is smaller than k_correspondences_
    // (gicp.hpp:72-79; k is lowered to |source| for sources under 20 points,
    // IcpLoopComputation.cc:487-489).  The reference then reads an empty covariance vector;
    // here such a pair ends unconverged.
    const int k_eff = pd.n_src < 20 ? pd.n_src : cfg.k_correspondences;
    if (k_eff > pd.n_src || k_eff > pd.n_tgt) {
      s.active = 0;
      s.failed = 1;
      std::memcpy(s.Tfinal, pd.guess, sizeof(s.Tfinal));
    }
  }

  // the arena was sized from the plan's estimate before anything was carved out of it: nothing
  // may have been handed out past its end
  if (ar.off > W->arena.cap)
    return fail(ctx, LGICP_E_NOMEM, "batch scratch estimate too small (%zu bytes needed, %zu reserved)",
                ar.off, W->arena.cap);

  // ---- block maps ----------------------------------------------------------------------------
  // Rebuilt whenever pairs drop out of the outer loop: the 1-NN grid covers the active pairs
  // only, and when few are left their targets are split over several workgroups so that the
  // launch still offers >= ~4 workgroups per CU.
  const int QB = ctx->nn_exhaustive ? nn_queries_per_block(ctx->nn_variant) : NNP_THREADS;
  std::vector<int4>& h_bmap_nn = W->v_bmap_nn;
  std::vector<int4>& h_bmap_red = W->v_bmap_red;
  std::vector<int> h_ints((size_t)2 * n_pairs + 4, 0);
  bool any_split = false;
  auto build_maps = [&](const std::vector<int>& act) {
    long long base_blocks = 0;
    for (int p : act) base_blocks += (h_pairs[p].n_src + QB - 1) / QB;
    int splits = 1;
    if (!ctx->nn_exhaustive)
      splits = 1;  // a wave of the pruned kernel sweeps the whole target index
    else if (cfg.reserved[1] > 0)
      splits = cfg.reserved[1];
    else
      while (base_blocks * splits < 1024 && splits < 64) splits *= 2;
    h_bmap_nn.clear();
    h_bmap_red.clear();
    any_split = false;
    for (int p : act) {
      const PairDev& pd = h_pairs[p];
      const int tiles = bp.clouds[bp.pairs[p].tgt_cloud].n_pad / NN_TILE;
      const int sp = std::max(1, std::min(splits, tiles));
      any_split |= sp > 1;
      const int qt = (pd.n_src + QB - 1) / QB;
      for (int t = 0; t < qt; ++t)
        for (int s = 0; s < sp; ++s) h_bmap_nn.push_back(make_int4(p, t, s, sp));
      for (int c = 0; c < h_ints[p]; ++c) h_bmap_red.push_back(make_int4(p, c, h_ints[p], 0));
    }
    xcd_permute(h_bmap_nn, W->v_permute);
  };
  std::vector<int> all_pairs((size_t)n_pairs);
  size_t max_nn_blocks = 0;
  double evals_per_pass = 0;
  for (int p = 0; p < n_pairs; ++p) {
    all_pairs[p] = p;
    const PairDev& pd = h_pairs[p];
    // slabs of RED_CHUNK_POINTS source points (a function of the pair alone, so that its sums do
    // not depend on the batch it travels in), one workgroup each: short chains keep the
    // stragglers' iterations short
    h_ints[p] = std::max(1, std::min(RED_MAX_CHUNKS, (pd.n_src + RED_CHUNK_POINTS - 1) / RED_CHUNK_POINTS));
    const int tiles = bp.clouds[bp.pairs[p].tgt_cloud].n_pad / NN_TILE;
    // the exhaustive kernel may split a query tile's target sweep over up to 64 blocks; the pruned
    // kernel has one block per 64 queries
    max_nn_blocks += (size_t)((pd.n_src + QB - 1) / QB) * (size_t)(ctx->nn_exhaustive ? std::min(64, tiles) : 1);
    evals_per_pass += (double)pd.n_src * (double)pd.n_tgt;
  }
  build_maps(all_pairs);

  // ---- one upload -----------------------------------------------------------------------------
  // Every host-built table goes through one pinned staging buffer and one copy; the live 1-NN
  // block map sits last because later, smaller maps are written over it.
  const int n_all_nn = (int)h_bmap_nn.size(), n_all_red = (int)h_bmap_red.size();
  const bool all_split = any_split;
  size_t par_bytes = 0;
  auto region = [&](size_t bytes) {
    const size_t off = par_bytes;
    par_bytes = (par_bytes + bytes + 255) & ~(size_t)255;
    return off;
  };
  const size_t o_clouds = region(sizeof(CloudDev) * n_clouds);
  const size_t o_pairs = region(sizeof(PairDev) * n_pairs);
  const size_t o_state = region(sizeof(PairState) * n_pairs);
  const size_t o_segs = region(sizeof(SegmentDev) * h_segs.size());
  const size_t o_isegs = region(sizeof(IndexSegDev) * h_isegs.size());
  const size_t o_ints = region(sizeof(int) * h_ints.size());
  const size_t o_nn_all = region(sizeof(int4) * h_bmap_nn.size());
  const size_t o_red_all = region(sizeof(int4) * h_bmap_red.size());
  const size_t o_red = region(sizeof(int4) * h_bmap_red.size());
  const size_t o_nn = region(sizeof(int4) * h_bmap_nn.size());
  const size_t used_bytes = par_bytes;
  par_bytes = o_nn + sizeof(int4) * std::max(max_nn_blocks, h_bmap_nn.size());
  HIP_TRY(ctx, W->d_params.reserve(par_bytes));
  if (W->h_params_cap < used_bytes) {
    if (W->h_params) (void)hipHostFree(W->h_params);
    W->h_params = nullptr;
    W->h_params_cap = 0;
    const size_t want = used_bytes + used_bytes / 4 + 4096;
    HIP_TRY(ctx, hipHostMalloc((void**)&W->h_params, want, hipHostMallocDefault));
    W->h_params_cap = want;
  }
  auto stage = [&](size_t off, const void* src, size_t bytes) {
    if (bytes) std::memcpy(W->h_params + off, src, bytes);
  };
  stage(o_clouds, h_clouds.data(), sizeof(CloudDev) * n_clouds);
  stage(o_pairs, h_pairs.data(), sizeof(PairDev) * n_pairs);
  stage(o_state, h_state.data(), sizeof(PairState) * n_pairs);
  stage(o_segs, h_segs.data(), sizeof(SegmentDev) * h_segs.size());
  stage(o_isegs, h_isegs.data(), sizeof(IndexSegDev) * h_isegs.size());
  stage(o_ints, h_ints.data(), sizeof(int) * h_ints.size());
  stage(o_nn_all, h_bmap_nn.data(), sizeof(int4) * h_bmap_nn.size());
  stage(o_red_all, h_bmap_red.data(), sizeof(int4) * h_bmap_red.size());
  stage(o_red, h_bmap_red.data(), sizeof(int4) * h_bmap_red.size());
  stage(o_nn, h_bmap_nn.data(), sizeof(int4) * h_bmap_nn.size());
  char* const dpar = W->d_params.p;
  CloudDev* const d_clouds = reinterpret_cast<CloudDev*>(dpar + o_clouds);
  PairDev* const d_pairs = reinterpret_cast<PairDev*>(dpar + o_pairs);
  PairState* const d_state = reinterpret_cast<PairState*>(dpar + o_state);
  SegmentDev* const d_segs = reinterpret_cast<SegmentDev*>(dpar + o_segs);
  IndexSegDev* const d_isegs = reinterpret_cast<IndexSegDev*>(dpar + o_isegs);
  int* const d_ints = reinterpret_cast<int*>(dpar + o_ints);
  int4* const d_bmap_nn_all = reinterpret_cast<int4*>(dpar + o_nn_all);
  int4* const d_bmap_red_all = reinterpret_cast<int4*>(dpar + o_red_all);
  int4* const d_bmap_red = reinterpret_cast<int4*>(dpar + o_red);
  int4* const d_bmap_nn = reinterpret_cast<int4*>(dpar + o_nn);
  HIP_TRY(ctx, W->d_partials.reserve((size_t)n_pairs * RED_MAX_CHUNKS * NSTAT));
  // the entry after the lookup counters is the solve kernels' arrival ticket (zero between launches)
  HIP_TRY(ctx, W->d_counters.reserve(NN_COUNTER_SLOTS + 1));
  HIP_TRY(ctx, hipMemsetAsync(W->d_counters.p, 0, sizeof(unsigned long long) * (NN_COUNTER_SLOTS + 1), st));
  int* const d_ticket = reinterpret_cast<int*>(W->d_counters.p + NN_COUNTER_SLOTS);
  // the staging buffer is rewritten only by the next batch of this slot, which starts after this
  // batch's last synchronisation
  HIP_TRY(ctx, launch_copy16(dpar, W->h_params, used_bytes, st));

  W->ev_used = 0;
  hipEvent_t ev_begin = W->event(), ev_prep = W->event();
  HIP_TRY(ctx, hipEventRecord(ev_begin, st));
  since("tables uploaded");

  // ---- clouds: accumulate, pad, covariances, normalisation constants --------------------------
  HIP_TRY(ctx, launch_assemble(d_segs, (int)h_segs.size(), max_seg_n, st));
  for (int c = 0; c < n_clouds; ++c)
    if (bp.clouds[c].in_arena())
      HIP_TRY(ctx, launch_pad_fill(const_cast<float4*>(h_clouds[c].xyz), h_clouds[c].n,
                                   h_clouds[c].n_pad, st));
  HIP_TRY(ctx, launch_index_fill(d_isegs, (int)h_isegs.size(), max_iseg_n, st));

  for (int c : normal_jobs)
    HIP_TRY(ctx, launch_normals_knn(h_clouds[c].xyz, h_clouds[c].n, h_clouds[c].n_pad,
                                    std::min(10, h_clouds[c].n),
                                    const_cast<float4*>(h_clouds[c].nrm_x), st));
  if (cfg.covariance_from_knn) {
    for (int c = 0; c < n_clouds; ++c)
      HIP_TRY(ctx, launch_cov_knn(h_clouds[c].xyz, h_clouds[c].n, h_clouds[c].n_pad,
                                  cfg.k_correspondences, cfg.gicp_epsilon,
                                  const_cast<double*>(h_clouds[c].cov), st));
  }
  // covariances from normals and the NormalizePCloud constants were derived when the keyframes
  // were registered; accumulated clouds received their members' covariances in launch_assemble
  // (with their flagged normals and 32-byte records); 64-byte records go with k-NN covariances
  if (cfg.covariance_from_knn)
    for (int c = 0; c < n_clouds; ++c)
      HIP_TRY(ctx, launch_pack_records(h_clouds[c].xyz, h_clouds[c].cov, h_clouds[c].n,
                                       const_cast<float4*>(h_clouds[c].rec), st));

  // ---- FEATURES / TEASERPP initialisation -----------------------------------------------------
  std::vector<char> init_rejected((size_t)n_pairs, 0);
  if (!explicit_guess &&
      (cfg.init_method == LGICP_INIT_FEATURES || cfg.init_method == LGICP_INIT_TEASERPP)) {
    const int rc = initialise_from_features(ctx, bp, h_clouds, cand, out, h_pairs, h_state, init_rejected, W->stats,
                                            (int)(W - ctx->slot) / FEATURE_LANES);
    if (rc != LGICP_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(d_pairs, h_pairs.data(), sizeof(PairDev) * n_pairs,
                                hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_state, h_state.data(), sizeof(PairState) * n_pairs,
                                hipMemcpyHostToDevice, st));
  }
  HIP_TRY(ctx, launch_guess_transform(d_clouds, d_pairs, n_pairs, max_src_n, st));
  // a pair that never enters the loop gets no lookup before the final one: its certificates must
  // read as "never certain" instead of whatever the arena held
  for (int p = 0; p < n_pairs; ++p)
    if (!h_state[p].active && h_pairs[p].cert)
      HIP_TRY(ctx, hipMemsetAsync(h_pairs[p].cert, 0, sizeof(float4) * (size_t)h_pairs[p].n_src, st));
  if (stage_events) HIP_TRY(ctx, hipEventRecord(ev_prep, st));

  // ---- GICP outer loop -----------------------------------------------------------------------
  SolveParams sp;
  sp.corr_dist2 = cfg.corr_dist * cfg.corr_dist;
  sp.rotation_epsilon = cfg.rotation_epsilon;
  sp.tf_epsilon = cfg.tf_epsilon;
  sp.max_iterations = cfg.iterations;
  sp.max_inner_iterations = cfg.optimizer_iterations;
  sp.cycle_exit = (cfg.reserved[5] & 2) ? 0 : 1;
  sp.pad = 0;
  // "Left the loop" flags, one per pair, raised by the solve kernel straight in pinned host memory
  // (the number of the iteration it happened in); entry 0 is the number of the last iteration whose
  // solve kernel has finished altogether.  No copy, no event and no second stream between the
  // device and the host's view: an event recorded between two kernels costs the stream about as
  // much as a straggler's lookup.
  int* const h_live = W->h_done;
  std::memset(h_live, 0, sizeof(int) * ((size_t)n_pairs + 1));
  int* d_done = nullptr;
  HIP_TRY(ctx, hipHostGetDevicePointer((void**)&d_done, h_live, 0));
  // One 1-NN pass over the pairs of the current block map.  Inside the GICP loop only matches
  // closer than the correspondence threshold are used (gicp.hpp:476), so the pruned search is
  // capped just above it; the pass after the loop is exact for every point.
  float nn_cap = std::nextafter((float)sp.corr_dist2, std::numeric_limits<float>::infinity());
  if (cfg.reserved[4] > 0)  // diagnostics only: a smaller cap drops correspondences
    nn_cap = 1e-6f * (float)cfg.reserved[4] * (float)cfg.reserved[4];
  bool have_matches = false;  // a previous pass has filled PairDev::keys: the next one may seed
  const bool lagged = !ctx->nn_exhaustive && !(cfg.reserved[5] & 1);
  int n_enqueued = 0;
  auto nn_pass = [&](bool final_pass) -> hipError_t {
    const bool seed = have_matches && cfg.reserved[6] != 1;
    have_matches = true;
    const int4* map = final_pass ? d_bmap_nn_all : d_bmap_nn;
    const int n_map = final_pass ? n_all_nn : (int)h_bmap_nn.size();
    const bool split = final_pass ? all_split : any_split;
    if (!ctx->nn_exhaustive)
      return launch_nn_pruned(d_clouds, d_pairs, d_state, map, n_map,
                              final_pass,
                              final_pass ? std::numeric_limits<float>::infinity() : nn_cap,
                              ctx->nn_r0sq, seed, W->d_counters.p, st);
    if (split) {
      hipError_t e = launch_keys_fill(d_pairs, d_state, n_pairs, max_src_n, final_pass, st);
      if (e != hipSuccess) return e;
    }
    return launch_nn_search(d_clouds, d_pairs, d_state, map, n_map,
                            final_pass, split, ctx->nn_variant, st);
  };
  std::vector<int> active;
  for (int p = 0; p < n_pairs; ++p)
    if (h_state[p].active) active.push_back(p);
  const int n_inactive0 = n_pairs - (int)active.size();
  // Live block maps are rewritten from one of two alternating pinned staging buffers with a
  // single copy (the reduce map's region is directly followed by the 1-NN map's).
  const size_t red_region = o_nn - o_red;
  int stage_k = 0;
  int stage_after[2] = {-1, -1};  // the loop launch a staging buffer's last copy was enqueued behind
  auto upload_maps = [&]() -> int {
    const size_t bytes = red_region + sizeof(int4) * h_bmap_nn.size();
    const int k = stage_k;
    stage_k ^= 1;
    if (W->h_maps_cap[k] < bytes) {
      if (W->h_maps[k]) (void)hipHostFree(W->h_maps[k]);
      W->h_maps[k] = nullptr;
      W->h_maps_cap[k] = 0;
      const size_t want = red_region + sizeof(int4) * std::max(max_nn_blocks, (size_t)n_all_nn) + 4096;
      HIP_TRY(ctx, hipHostMalloc((void**)&W->h_maps[k], want, hipHostMallocDefault));
      W->h_maps_cap[k] = want;
    }
    char* h = reinterpret_cast<char*>(W->h_maps[k]);
    std::memcpy(h, h_bmap_red.data(), sizeof(int4) * h_bmap_red.size());
    std::memcpy(h + red_region, h_bmap_nn.data(), sizeof(int4) * h_bmap_nn.size());
    HIP_TRY(ctx, launch_copy16(d_bmap_red, h, bytes, st));
    return LGICP_OK;
  };
  if (n_inactive0) {
    build_maps(active);
    stage_after[stage_k] = 0;
    const int rc = upload_maps();
    if (rc != LGICP_OK) return rc;
  }
  size_t mapped_active = active.size();
  std::vector<std::pair<hipEvent_t, hipEvent_t>> nn_ev, red_ev, sol_ev;
  std::vector<int> h_flags((size_t)n_pairs + 1, 0);
  int done_seen = 0;
  // The host stays one iteration behind the device: iteration i+1 is enqueued before the
  // "done" flags of iteration i are looked at, so the device never waits for the round trip.
  // Pairs that have left the loop are skipped by the kernels themselves (PairState::active);
  // the block maps only follow, two iterations late, to keep the grids small.
  auto enqueue_iteration = [&]() -> int {
    hipEvent_t a = nullptr, b = nullptr, c = nullptr, d = nullptr;
    if (stage_events) a = W->event(), b = W->event(), c = W->event(), d = W->event();
    const int iter = ++n_enqueued;
    if (stage_events) HIP_TRY(ctx, hipEventRecord(a, st));
    HIP_TRY(ctx, nn_pass(false));
    if (stage_events) HIP_TRY(ctx, hipEventRecord(b, st));
    if (cfg.solver == LGICP_SOLVER_BFGS) {
      HIP_TRY(ctx, launch_bfgs_prepare(d_clouds, d_pairs, d_state,
                                       d_bmap_red, (int)h_bmap_red.size(), sp.corr_dist2,
                                       W->d_partials.p, !cfg.covariance_from_knn, st));
      if (stage_events) HIP_TRY(ctx, hipEventRecord(c, st));
      HIP_TRY(ctx, launch_gicp_bfgs(d_pairs, d_state, n_pairs, d_ints,
                                    W->d_partials.p, sp, d_done, iter, d_ticket, st));
    } else {
      HIP_TRY(ctx, launch_gicp_reduce(d_clouds, d_pairs, d_state,
                                      d_bmap_red, (int)h_bmap_red.size(), sp.corr_dist2,
                                      W->d_partials.p, !cfg.covariance_from_knn, st));
      if (stage_events) HIP_TRY(ctx, hipEventRecord(c, st));
      HIP_TRY(ctx, launch_gicp_solve(d_pairs, d_state, n_pairs, d_ints,
                                     W->d_partials.p, sp, d_done, iter, d_ticket, st));
    }
    if (stage_events) {
      HIP_TRY(ctx, hipEventRecord(d, st));
      nn_ev.emplace_back(a, b);
      red_ev.emplace_back(b, c);
      sol_ev.emplace_back(c, d);
    }
    return LGICP_OK;
  };
  // Waits until the solve kernel of iteration k has finished (every flag of iterations <= k is then
  // visible; the solve kernel of a later iteration that is already running can only add flags).
  // The stream is looked at now and then: a launch that failed would never report.
  auto wait_iteration = [&](int k) -> hipError_t {
    const volatile int* const it = h_live;
    for (unsigned spins = 1;; ++spins) {
      if (it[0] >= k) return hipSuccess;
      if ((spins & 1023u) == 0) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return it[0] >= k ? hipSuccess : hipErrorUnknown;
        if (e != hipErrorNotReady) return e;
      }
      __builtin_ia32_pause();
    }
  };
  const int n_active0 = (int)active.size();
  // launches of the loop: the kernels enforce the iteration limit themselves, one more launch finds
  // every pair stopped
  const int max_launches = cfg.iterations + 1;
  // The host stays one iteration behind the device: iteration i + 1 is enqueued before the flags
  // of iteration i are looked at (two or three ahead measured no faster: the host keeps up).
  const int ahead = lagged ? 1 : 0;
  int completed = 0;  // iterations whose solve kernel is known to be through
  if (!active.empty()) {
    for (;;) {
      while (n_enqueued < std::min(completed + 1 + ahead, max_launches)) {
        const int rc = enqueue_iteration();
        if (rc != LGICP_OK) return rc;
      }
      HIP_TRY(ctx, wait_iteration(completed + 1));
      ++completed;
      const volatile int* flags = h_live;
      // (a flag of an iteration that is running now may be up already)
      int n_flagged = 0;
      for (int p = 0; p < n_pairs; ++p) n_flagged += flags[1 + p] != 0;
      if (n_flagged >= n_active0 || completed >= max_launches) {
        if (n_enqueued > completed) HIP_TRY(ctx, hipStreamSynchronize(st));  // kernels found nothing to do
        break;
      }
      if (n_flagged != done_seen) {
        done_seen = n_flagged;
        active.clear();
        for (int p = 0; p < n_pairs; ++p)
          if (h_state[p].active && !flags[1 + p]) active.push_back(p);
        // Small changes are not worth a copy: blocks of finished pairs exit at once anyway.  A
        // staging buffer is reused two uploads later: only once the iteration enqueued behind its
        // copy is through.
        if (!active.empty() && active.size() * 8 <= mapped_active * 7 && completed > stage_after[stage_k]) {
          build_maps(active);
          stage_after[stage_k] = n_enqueued;
          const int rc = upload_maps();
          if (rc != LGICP_OK) return rc;
          mapped_active = active.size();
        }
      }
    }
  }
  since("loop done");

  // ---- final pass: fitness + correspondences + Ap ----------------------------------------------
  hipEvent_t ev_post0 = W->event(), ev_post1 = W->event(), ev_post2 = W->event();
  if (stage_events) HIP_TRY(ctx, hipEventRecord(ev_post0, st));
  HIP_TRY(ctx, nn_pass(true));
  if (stage_events) HIP_TRY(ctx, hipEventRecord(ev_post1, st));
  HIP_TRY(ctx, launch_post_reduce(d_clouds, d_pairs, d_state,
                                  d_bmap_red_all, n_all_red, W->d_partials.p,
                                  cfg.covariance_method == LGICP_COV_POINT2POINT &&
                                      !cfg.use_fixed_covariances,
                                  st));
  HIP_TRY(ctx, hipEventRecord(ev_post2, st));
  // downloads land in pinned memory (a pageable destination makes every copy a staged,
  // synchronous one)
  const size_t post_bytes = sizeof(double) * (size_t)n_pairs * RED_MAX_CHUNKS * NPOST;
  const size_t state_bytes = sizeof(PairState) * (size_t)n_pairs;
  const size_t counter_bytes = sizeof(unsigned long long) * NN_COUNTER_SLOTS;
  const size_t o_hstate = (post_bytes + 255) & ~(size_t)255;
  const size_t o_hcount = (o_hstate + state_bytes + 255) & ~(size_t)255;
  if (W->h_results_cap < o_hcount + counter_bytes) {
    if (W->h_results) (void)hipHostFree(W->h_results);
    W->h_results = nullptr, W->h_results_cap = 0;
    const size_t want = (o_hcount + counter_bytes) * 5 / 4 + 4096;
    HIP_TRY(ctx, hipHostMalloc((void**)&W->h_results, want, hipHostMallocDefault));
    W->h_results_cap = want;
  }
  const double* const h_post = reinterpret_cast<const double*>(W->h_results);
  const unsigned long long* const h_counters =
      reinterpret_cast<const unsigned long long*>(W->h_results + o_hcount);
  HIP_TRY(ctx, launch_copy16(W->h_results, W->d_partials.p, post_bytes, st));
  HIP_TRY(ctx, launch_copy16(W->h_results + o_hstate, d_state, state_bytes, st));
  HIP_TRY(ctx, launch_copy16(W->h_results + o_hcount, W->d_counters.p, counter_bytes, st));
  {
    hipEvent_t ev_done = W->event();
    HIP_TRY(ctx, hipEventRecord(ev_done, st));
    HIP_TRY(ctx, spin_until(ev_done));
  }
  std::memcpy(h_state.data(), W->h_results + o_hstate, state_bytes);
  since("results downloaded");
  for (int k = 0; k < NN_COUNTER_SLOTS; ++k) W->stats.nn_executed_evals += (double)h_counters[k];

  // ---- timing ------------------------------------------------------------------------------
  auto sum_ms = [](const std::vector<std::pair<hipEvent_t, hipEvent_t>>& v) {
    double s = 0;
    for (auto& e : v) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) s += ms;
    }
    return s;
  };
  float ms = 0;
  lgicp_stats& S = W->stats;
  if (trace_t) {  // kernel times of every outer iteration, in microseconds
    std::string line = "lgicp trace: nn/reduce/solve us per iteration:";
    char buf[64];
    for (size_t k = 0; k < nn_ev.size(); ++k) {
      float a = 0, b = 0, c = 0;
      (void)hipEventElapsedTime(&a, nn_ev[k].first, nn_ev[k].second);
      (void)hipEventElapsedTime(&b, red_ev[k].first, red_ev[k].second);
      (void)hipEventElapsedTime(&c, sol_ev[k].first, sol_ev[k].second);
      snprintf(buf, sizeof(buf), " %.0f/%.0f/%.0f", 1e3 * a, 1e3 * b, 1e3 * c);
      line += buf;
    }
    fprintf(stderr, "%s\n", line.c_str());
  }
  // per-stage times only with lgicp_config::stage_timing (the events they need sit between the
  // kernels of every iteration)
  double nn_final_ms = 0;
  if (stage_events) {
    (void)hipEventElapsedTime(&ms, ev_post0, ev_post1);
    nn_final_ms = ms;
    S.nn_ms += sum_ms(nn_ev) + nn_final_ms;
    S.reduce_ms += sum_ms(red_ev);
    S.solve_ms += sum_ms(sol_ev);
    (void)hipEventElapsedTime(&ms, ev_begin, ev_prep);
    S.cov_ms += ms;
    (void)hipEventElapsedTime(&ms, ev_post1, ev_post2);
    S.post_ms += ms;
  }
  (void)hipEventElapsedTime(&ms, ev_begin, ev_post2);
  S.total_ms += ms;
  S.nn_launches += (uint64_t)n_enqueued + 1;
  S.reduce_launches += (uint64_t)n_enqueued;
  // the final pass always covers every pair of the batch: it is the launch the roofline uses
  S.last_nn_kernel_ms = nn_final_ms;
  S.last_nn_evals_per_launch = evals_per_pass;

  finish_pairs(cfg, bp, h_pairs, h_state, h_ints, h_post, init_rejected, out, S);
  since("batch done");
  return LGICP_OK;
}

// explicit_guess: the pairs' guesses are the caller's and go to GICP as they are, whatever the
// context's initialisation method says (lgicp_align).  On an error the slot's stream is drained
// before returning: kernels already queued still write the pinned flags and the arena, which the
// next batch of the slot clears and reuses.
int run_batch(lgicp_ctx* ctx, BatchSlot* W, BatchPlan& bp, const lgicp_candidate* cand,
              lgicp_result* out, bool explicit_guess = false) {
  const int rc = run_batch_impl(ctx, W, bp, cand, out, explicit_guess);
  if (rc != LGICP_OK) (void)hipStreamSynchronize(W->stream);
  return rc;
}

}  // namespace

namespace {

void merge_stats(lgicp_stats& into, const lgicp_stats& d) {
  into.pairs_ok += d.pairs_ok;
  into.outer_iterations += d.outer_iterations;
  into.nn_launches += d.nn_launches;
  into.nn_distance_evals += d.nn_distance_evals;
  into.nn_ms += d.nn_ms, into.reduce_ms += d.reduce_ms, into.solve_ms += d.solve_ms;
  into.cov_ms += d.cov_ms, into.post_ms += d.post_ms;
  into.nn_executed_evals += d.nn_executed_evals;
  into.reduce_points += d.reduce_points, into.reduce_inliers += d.reduce_inliers;
  into.reduce_launches += d.reduce_launches;
  into.cycle_exits += d.cycle_exits;
  into.clique_ms += d.clique_ms, into.clique_row_bytes += d.clique_row_bytes;
  into.clique_launches += d.clique_launches;
  into.core_ms += d.core_ms, into.core_matrix_bytes += d.core_matrix_bytes;
  into.core_launches += d.core_launches;
  into.teaser_oversized_pairs += d.teaser_oversized_pairs;
  if (d.last_nn_evals_per_launch > 0) {
    into.last_nn_kernel_ms = d.last_nn_kernel_ms;
    into.last_nn_evals_per_launch = d.last_nn_evals_per_launch;
  }
}

void reset_submit_stats(lgicp_stats& s) {
  s.nn_ms = s.reduce_ms = s.solve_ms = s.cov_ms = s.post_ms = s.total_ms = 0;
  s.nn_executed_evals = s.reduce_points = s.reduce_inliers = 0;
  s.reduce_launches = 0;
  s.cycle_exits = 0;
  s.clique_ms = s.clique_row_bytes = s.core_ms = s.core_matrix_bytes = 0;
  s.clique_launches = s.core_launches = 0;
  s.teaser_oversized_pairs = 0;
}

}  // namespace

extern "C" int lgicp_submit(lgicp_ctx* ctx, const lgicp_candidate* cand, size_t n,
                            lgicp_result* out) {
  return lgicp_submit_stats(ctx, cand, n, out, nullptr);
}

extern "C" int lgicp_submit_stats(lgicp_ctx* ctx, const lgicp_candidate* cand, size_t n,
                                  lgicp_result* out, lgicp_stats* submit_stats) {
  if (submit_stats) std::memset(submit_stats, 0, sizeof(*submit_stats));
  if (!ctx) return LGICP_E_INVALID;
  if (n && (!cand || !out)) return fail(ctx, LGICP_E_INVALID, "null argument");
  const lgicp_config& cfg = ctx->cfg;
  const bool trace_submit = std::getenv("LGICP_TRACE") != nullptr;
  const auto t_submit = std::chrono::steady_clock::now();
  auto submit_mark = [&](const char* what) {
    if (trace_submit)
      std::fprintf(stderr, "[lgicp] submit %s at %.3f ms\n", what,
                   1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_submit).count());
  };
  HIP_TRY(ctx, hipSetDevice(cfg.device));
  // Keyframes and poses stay put for the duration (registration waits).  A submit runs on one lane
  // of the context.
  std::shared_lock<std::shared_mutex> data(ctx->data_mu, std::defer_lock);
  {
    std::lock_guard<std::mutex> gate(ctx->gate_mu);  // behind any registration that is waiting
    data.lock();
  }
  // (a context with a feature-based initialisation has at most two lanes, each with one of the
  // context's two sets of feature buffers and four streams, see lgicp_set_submit_lanes)
  const bool feature_init = cfg.init_method == LGICP_INIT_FEATURES || cfg.init_method == LGICP_INIT_TEASERPP;
  const bool whole_context = ctx->lanes.lanes() == 1;
  // slots per lane, and the lanes that cover the slots one submit can be spread over
  const int per_lane = feature_init ? FEATURE_LANES : std::max(1, ctx->n_slots / ctx->lanes.lanes());
  const int wide_lanes = feature_init ? 1 : (ctx->n_wide + per_lane - 1) / per_lane;
  std::unique_ptr<AllLanes> hold_all(whole_context ? new AllLanes(ctx) : nullptr);
  std::unique_ptr<OneLane> hold_one(whole_context ? nullptr : new OneLane(ctx, wide_lanes));
  const int lane = hold_one ? hold_one->lane : 0;
  const int lane_slots = hold_one ? std::min(per_lane * hold_one->held, std::max(per_lane, ctx->n_wide))
                                  : ctx->n_wide;
  const int first_slot = lane * per_lane;
  // the helper threads of the first lane held run the other parts of this submit
  const int helper_slots = (int)ctx->helpers[lane].threads.size() + 1;
  {
    // keyframes registered since the last submit: their device work runs on the context's stream
    std::lock_guard<std::mutex> fl(ctx->flush_mu);
    const int rc = flush_registrations(ctx);
    if (rc != LGICP_OK) return rc;
  }
  lgicp_stats sub;  // this submit's share of the statistics
  std::memset(&sub, 0, sizeof(sub));

  // PerformAlignment's availability checks (IcpLoopComputation.cc:438-465)
  size_t n_valid = 0;
  for (size_t i = 0; i < n; ++i) {
    const lgicp_candidate& c = cand[i];
    lgicp_result& r = out[i];
    std::memset(&r, 0, sizeof(r));
    r.key_from = c.key_from;
    r.key_to = c.key_to;
    r.status = LGICP_PAIR_OK;
    sub.pairs_submitted++;
    if (!ctx->scans.count(c.key_from) || !ctx->scans.count(c.key_to))
      r.status = LGICP_PAIR_MISSING_SCAN;
    else if (!ctx->poses.count(c.key_from) || !ctx->poses.count(c.key_to))
      r.status = LGICP_PAIR_MISSING_POSE;
    else if (ctx->scans.at(c.key_from).n == 0 || ctx->scans.at(c.key_to).n == 0)
      r.status = LGICP_PAIR_EMPTY_CLOUD;
    else
      ++n_valid;
  }

  // Several parts in flight at once when the submit is large enough (the feature-based
  // initialisations run their host stage on the context's own stream and stay serial).
  int ways = std::min(lane_slots, helper_slots);
  if (cfg.init_method == LGICP_INIT_FEATURES || cfg.init_method == LGICP_INIT_TEASERPP) ways = 1;
  while (ways > 1 && n_valid < (size_t)32 * ways) --ways;
  const size_t max_pairs = std::min<size_t>(32767, (n_valid + ways - 1) / std::max(ways, 1));
  const size_t max_bytes = cfg.workspace_bytes / ways;

  std::atomic<int> rc_all{LGICP_OK};
  std::mutex stats_mu;
  double slot_total_ms[lgicp_ctx::MAX_SLOTS] = {};
  std::vector<BatchPlan> plans;
  std::atomic<bool> plans_ready{false};
  auto worker = [&](int si) {
    // helpers are woken before the plans exist, so that the wake-up overlaps the planning
    while (!plans_ready.load(std::memory_order_acquire)) std::this_thread::yield();
    if (hipSetDevice(cfg.device) != hipSuccess) {
      rc_all = fail(ctx, LGICP_E_HIP, "hipSetDevice failed");
      return;
    }
    BatchSlot* W = &ctx->slot[first_slot + si];
    // part k of a submit always runs in slot k mod ways: repeated submits of the same shape find
    // every slot's buffers already sized for their part
    for (size_t k = (size_t)si; k < plans.size() && rc_all == LGICP_OK; k += (size_t)ways) {
      std::memset(&W->stats, 0, sizeof(W->stats));
      const int rc = run_batch(ctx, W, plans[k], cand, out);
      std::lock_guard<std::mutex> lock(stats_mu);
      merge_stats(sub, W->stats);
      slot_total_ms[si] += W->stats.total_ms;
      if (rc != LGICP_OK) {
        rc_all = rc;
        break;
      }
    }
  };
  ctx->helpers[lane].post(ways - 1, worker);
  size_t i = 0;
  while (i < n) {
    BatchPlan bp;
    size_t pair_bytes = 0;
    for (; i < n; ++i) {
      const lgicp_candidate& c = cand[i];
      if (out[i].status != LGICP_PAIR_OK) continue;
      const size_t est = (size_t)ctx->scans.at(c.key_from).n * 24 * (cfg.accumulate_source ? 5 : 1);
      if (!bp.pairs.empty() &&
          (bp.bytes + pair_bytes + est > max_bytes || bp.pairs.size() >= max_pairs))
        break;
      PairPlan pp;
      pp.cand = i;
      pp.tgt_cloud = plan_cloud(ctx, bp, c.key_to, true);
      pp.src_cloud = plan_cloud(ctx, bp, c.key_from, cfg.accumulate_source != 0);
      pp.scan1_cloud = plan_cloud(ctx, bp, c.key_from, false);
      pp.scan2_cloud = plan_cloud(ctx, bp, c.key_to, false);
      bp.clouds[pp.scan1_cloud].need_norm_constants = true;
      // initial guess (IcpLoopComputation.cc:495-543)
      pp.pose21_odom = hm::between(ctx->poses.at(c.key_to), ctx->poses.at(c.key_from));
      switch (cfg.init_method) {
        case LGICP_INIT_ODOMETRY:
        case LGICP_INIT_FEATURES:  // SAC-IA and TEASER++ start from the odometry guess
        case LGICP_INIT_TEASERPP:
          guess_from_pose(pp.pose21_odom, true, pp.guess);
          break;
        case LGICP_INIT_ODOM_ROTATION: guess_from_pose(pp.pose21_odom, false, pp.guess); break;
        case LGICP_INIT_CANDIDATE: {
          const hm::Pose p1 = hm::pose_from_quat(c.pose_from_q, c.pose_from_t);
          const hm::Pose p2 = hm::pose_from_quat(c.pose_to_q, c.pose_to_t);
          guess_from_pose(hm::between(p2, p1), true, pp.guess);
        } break;
        default:
          for (int k = 0; k < 16; ++k) pp.guess[k] = (k % 5 == 0) ? 1.f : 0.f;
      }
      pair_bytes += pair_scratch_bytes(cfg, bp.clouds[pp.src_cloud].n);
      bp.pairs.push_back(pp);
    }
    if (!bp.pairs.empty()) plans.push_back(std::move(bp));
  }

  submit_mark("planned");
  {
    const int rc = resolve_accumulated(ctx, plans, &ctx->slot[first_slot]);
    if (rc != LGICP_OK) {  // nothing of the submit runs
      rc_all = rc;
      plans.clear();
    }
  }
  plans_ready.store(true, std::memory_order_release);
  worker(0);
  submit_mark("own part done");
  ctx->helpers[lane].wait();
  submit_mark("all parts done");
  // device span of the submit: the slots overlap
  sub.total_ms = *std::max_element(slot_total_ms, slot_total_ms + lgicp_ctx::MAX_SLOTS);
  {
    // cumulative fields grow, the "last submit" fields become this submit's
    std::lock_guard<std::mutex> lk(ctx->stats_mu);
    reset_submit_stats(ctx->stats);
    merge_stats(ctx->stats, sub);
    ctx->stats.pairs_submitted += sub.pairs_submitted;
    ctx->stats.total_ms = sub.total_ms;
  }
  if (submit_stats) *submit_stats = sub;
  return rc_all;
}

// One alignment with an explicit guess: the two clouds become temporary keyframes (keys far
// from any neighbour, so nothing is accumulated) and go through the same batch path.
extern "C" int lgicp_align(lgicp_ctx* ctx, const float* src_xyz, const float* src_normals,
                           size_t n_src, const float* tgt_xyz, const float* tgt_normals,
                           size_t n_tgt, const float guess[16], float final_T[16], int* converged,
                           int* iterations, double* fitness) {
  if (!ctx || !guess || !final_T) return fail(ctx, LGICP_E_INVALID, "null argument");
  if (n_src == 0 || n_tgt == 0) return fail(ctx, LGICP_E_INVALID, "empty cloud");
  const uint64_t ks = 0xFFFFFFF000000000ull, kt = 0xFFFFFFE000000000ull;
  std::lock_guard<std::mutex> gate(ctx->gate_mu);
  std::unique_lock<std::shared_mutex> data(ctx->data_mu);  // adds and removes two keyframes
  AllLanes all(ctx);
  if (ctx->scans.count(ks) || ctx->scans.count(kt))
    return fail(ctx, LGICP_E_INVALID, "reserved keys in use");
  if (hipSetDevice(ctx->cfg.device) != hipSuccess) return fail(ctx, LGICP_E_HIP, "hipSetDevice failed");
  int rc = add_keyframes_locked(ctx, 1, &ks, &src_xyz, src_normals ? &src_normals : nullptr, &n_src, 12);
  if (rc < 0) return rc;
  rc = add_keyframes_locked(ctx, 1, &kt, &tgt_xyz, tgt_normals ? &tgt_normals : nullptr, &n_tgt, 12);
  if (rc >= 0) rc = flush_registrations(ctx);
  if (rc >= 0) {
    const double q[4] = {1, 0, 0, 0}, t[3] = {0, 0, 0};
    ctx->poses.emplace(ks, hm::pose_from_quat(q, t));
    ctx->poses.emplace(kt, hm::pose_from_quat(q, t));
    BatchPlan bp;
    PairPlan pp;
    pp.cand = 0;
    pp.tgt_cloud = plan_cloud(ctx, bp, kt, false);
    pp.src_cloud = plan_cloud(ctx, bp, ks, false);
    pp.scan1_cloud = pp.src_cloud;
    pp.scan2_cloud = pp.tgt_cloud;
    bp.clouds[pp.scan1_cloud].need_norm_constants = true;
    pp.pose21_odom = hm::pose_from_quat(q, t);
    std::memcpy(pp.guess, guess, sizeof(pp.guess));
    bp.pairs.push_back(pp);
    lgicp_candidate c;
    std::memset(&c, 0, sizeof(c));
    lgicp_result r;
    std::memset(&r, 0, sizeof(r));
    std::memset(&ctx->slot[0].stats, 0, sizeof(lgicp_stats));
    rc = run_batch(ctx, &ctx->slot[0], bp, &c, &r, /*explicit_guess=*/true);
    {
      std::lock_guard<std::mutex> lk(ctx->stats_mu);
      reset_submit_stats(ctx->stats);
      merge_stats(ctx->stats, ctx->slot[0].stats);
      ctx->stats.total_ms = ctx->slot[0].stats.total_ms;
    }
    if (rc == LGICP_OK) {
      std::memcpy(final_T, r.T_icp, sizeof(r.T_icp));
      if (converged) *converged = r.converged;
      if (iterations) *iterations = r.iterations;
      if (fitness) *fitness = r.fitness;
    }
  }
  for (uint64_t k : {ks, kt}) {
    auto it = ctx->scans.find(k);
    if (it != ctx->scans.end()) {
      it->second.free_device();
      ctx->scans.erase(it);
    }
    ctx->poses.erase(k);
  }
  return rc < 0 ? rc : LGICP_OK;
}
