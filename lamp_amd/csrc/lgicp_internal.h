// lgicp_internal.h — context object and helpers shared by the host translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <functional>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/lamp_gicp.h"
#include "host_math.h"
#include "lane_table.h"
#include "lgicp_device.h"
#include "teaser.h"

using namespace lgicp;

struct lgicp_ctx;

namespace lgicp_detail {

inline std::string g_create_error;

struct CloudFeatures;
struct CliqueBuffers;
struct TeaserJob;
struct MatcherDev;
struct FeatureScratch;
struct SacLane;

struct DevBlock {  // one device allocation shared by the keyframes registered in one call
  char* p = nullptr;
  ~DevBlock() {
    if (p) (void)hipFree(p);
  }
};

struct Keyframe {
  int n = 0;
  int n_pad = 0;
  int nf = 0;  // points with finite coordinates
  std::shared_ptr<DevBlock> block;  // owns every d_* array below
  float4* d_xyz = nullptr;  // n_pad
  float4* d_nrm = nullptr;  // n
  // spatial index of the keyframe on its own (see CloudDev)
  int n_tiles = 0;
  int n_q = 0;
  int* d_order = nullptr;   // n: visiting order (k-d partition)
  float4* d_sxyz = nullptr;
  float4* d_tmin = nullptr;
  float4* d_tmax = nullptr;
  int* d_qorder = nullptr;
  // derived once at registration: covariances from the normals (gicp_utils.h:28-62) and the
  // NormalizePCloud constants (PointCloudUtils.cc:39-59); both depend on the keyframe alone
  double* d_cov = nullptr;  // n x 6
  float4* d_rec = nullptr;  // n x 32 bytes: point + flagged normal, what the reduction gathers per match
  float4* d_cnrm = nullptr; // n: the stored normal, w = 1 where the covariance has to be read from d_cov
  // lamp_utils::ExtractNormals of the keyframe: d_nrm itself, or k-NN normals (own buffer) when
  // the first stored normal is all zero
  float4* d_nrm_x = nullptr;
  bool normals_missing = false;
  float centroid[3] = {0.f, 0.f, 0.f};
  float factor = 0.f;
  // Harris keypoints and FPFH descriptors of the keyframe on its own, computed by the first
  // FEATURES / TEASERPP submit that needs them (they depend on the keyframe and the context's
  // parameters alone, and a keyframe shows up in many candidates over its lifetime)
  std::shared_ptr<CloudFeatures> features;
  void free_device() {
    block.reset();
    d_xyz = d_nrm = d_sxyz = d_tmin = d_tmax = d_nrm_x = d_rec = d_cnrm = nullptr;
    d_order = d_qorder = nullptr;
    d_cov = nullptr;
  }
};

// hipFree waits for all work of the device, whichever stream it is on: a buffer that is outgrown
// in the middle of a submit (the groups of a feature-based initialisation differ from batch to
// batch) would stall every stage running beside the caller, once per buffer.  Outgrown buffers are
// parked instead -- with the context the parking thread is working for (ParkScope) -- and freed when
// a later submit of THAT context begins with nothing of the context in flight
// (free_parked_device_memory(ctx)): another context of the process may be in the middle of a
// submit then, and its parked buffers may still be read by its kernels.  A thread that works for
// no context parks in a process-wide list, freed by whichever context is destroyed next.  All
// three are safe to call from any thread.
void park_device_memory(void* p);
void park_pinned_memory(void* p);  // same for hipHostMalloc'ed memory
void free_parked_device_memory(struct ::lgicp_ctx* ctx);  // ctx == nullptr: the process-wide list only
extern thread_local struct ::lgicp_ctx* t_park_ctx;
struct ParkScope {  // the calling thread works for `c` while this exists
  struct ::lgicp_ctx* prev;
  explicit ParkScope(struct ::lgicp_ctx* c) : prev(t_park_ctx) { t_park_ctx = c; }
  ~ParkScope() { t_park_ctx = prev; }
  ParkScope(const ParkScope&) = delete;
  ParkScope& operator=(const ParkScope&) = delete;
};

template <typename T>
struct DevBuf {  // grow-only device array
  T* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    park_device_memory(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 4 + 16;
    hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// One cloud handed to register_keyframes (host pointers, `stride_floats` floats between points;
// nrm may be null: all-zero normals)
struct KfInput {
  uint64_t key;
  const float* xyz;
  const float* nrm;
  size_t n;
  size_t stride_floats;
};

// Buffers of keyframe registration, kept between calls (grow-only)
struct Registry {
  char* h_stage = nullptr;   // pinned: float4 points and normals of a batch on their way up
  size_t stage_cap = 0;
  char* h_tab = nullptr;     // pinned: per-keyframe tables of a batch
  size_t tab_cap = 0;
  char* h_const = nullptr;   // pinned, device-visible: CloudDev entries the NormalizePCloud kernel
  size_t const_cap = 0;      // writes centroid / factor into, followed by their ids
  size_t const_used = 0;
  hipEvent_t ev_stage = nullptr;  // the last batch's uploads have left h_stage / h_tab
  DevBuf<unsigned long long> d_keys[2];
  DevBuf<int> d_ids[2];
  DevBuf<unsigned> d_box;
  DevBuf<char> d_tmp, d_tab;
  struct Pending {
    uint64_t key;
    size_t slot;
  };
  std::vector<Pending> pending;  // keyframes whose constants have not been picked up yet
  void destroy();
};

// An accumulated cloud (IcpLoopComputation::AccumulateScans, IcpLoopComputation.cc:910-947: a
// keyframe plus its neighbours key +- i moved into its frame) kept on the device with everything a
// registered keyframe has: scans and poses never change once stored (first wins,
// IcpLoopComputation.cc:380-385, 400-402), so the cloud is a function of the keys it is made of.
// The reference rebuilds it, and the kd-tree over it, inside every PerformAlignment.
struct AccEntry {
  std::vector<uint64_t> members;  // the neighbours that went in, in AccumulateScans' order
  Keyframe kf;                    // points, normals, k-d index, covariances, records of the whole
  size_t bytes = 0;
  uint64_t last_use = 0;
};
// one accumulated cloud to build: the key's own keyframe, then (neighbour, its pose in the key's frame)
struct AccBuild {
  const Keyframe* own;
  std::vector<std::pair<const Keyframe*, hm::Pose>> segs;
  AccEntry* out;
};
// Builds out->kf of every entry on the device (one allocation each, freed through
// park_device_memory): assemble, pad, k-d index of the whole cloud, covariances / flagged normals /
// records from the copied normals.  Enqueued on `st` with the buffers of `R`; one caller per R at a
// time.
int register_accumulated(lgicp_ctx* ctx, Registry& R, hipStream_t st, AccBuild* builds, size_t count);

}  // namespace lgicp_detail
using namespace lgicp_detail;

// Everything one in-flight batch needs on the device and for talking to it.  A context owns
// several of these so that parts of a submit can be in flight at once: while one part sits in
// a latency-bound stretch (the 6x6 solves, the last stragglers of the outer loop, host
// bookkeeping) the other parts' bulk kernels keep the GPU busy.
struct BatchSlot {
  hipStream_t stream = nullptr;
  DevBuf<char> arena;                 // per-batch scratch (bump allocated)
  // clouds, pairs, states, segments, chunk counts + done flags, block maps: one buffer, one copy
  DevBuf<char> d_params;
  char* h_params = nullptr;  // pinned staging of the same
  size_t h_params_cap = 0;
  int4* h_maps[2] = {nullptr, nullptr};  // pinned staging of the block maps rebuilt in the loop
  size_t h_maps_cap[2] = {0, 0};
  DevBuf<double> d_partials;
  DevBuf<unsigned long long> d_counters;
  int* h_done = nullptr;  // pinned and device-visible: the solve kernel raises per-pair "done" flags here
  char* h_results = nullptr;  // pinned landing area of a batch's downloads (partials, states, counters)
  size_t h_results_cap = 0;
  // host tables of the batch in flight (kept between batches: no page faults on reuse)
  std::vector<lgicp::PairDev> v_pairs;
  std::vector<lgicp::PairState> v_state;
  std::vector<int4> v_bmap_nn, v_bmap_red, v_permute;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  lgicp_stats stats;  // what the batch in this slot adds to the context's statistics
  Registry acc_reg;   // buffers for building accumulated clouds on this slot's stream
  std::vector<std::shared_ptr<AccEntry>> acc_in_use;  // alive until the slot's next batch begins

  hipEvent_t event() {
    if (ev_used == ev_pool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      ev_pool.push_back(e);
    }
    return ev_pool[ev_used++];
  }
  // takes ownership of `s`
  bool create(hipStream_t s) {
    stream = s;
    return s != nullptr &&
           hipHostMalloc((void**)&h_done, sizeof(int) * 65536, hipHostMallocDefault) == hipSuccess;
  }
  void destroy() {
    if (stream) (void)hipStreamSynchronize(stream);
    acc_in_use.clear();
    acc_reg.destroy();
    arena.release(), d_params.release(), d_partials.release(), d_counters.release();
    if (h_params) (void)hipHostFree(h_params);
    for (int k = 0; k < 2; ++k)
      if (h_maps[k]) (void)hipHostFree(h_maps[k]);
    h_params = nullptr, h_maps[0] = h_maps[1] = nullptr;
    h_params_cap = h_maps_cap[0] = h_maps_cap[1] = 0;
    for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
    ev_pool.clear();
    if (h_done) (void)hipHostFree(h_done);
    if (h_results) (void)hipHostFree(h_results);
    h_results = nullptr, h_results_cap = 0;
    if (stream) (void)hipStreamDestroy(stream);
    h_done = nullptr, stream = nullptr;
  }
};

// Helper threads of a context, one per batch slot beyond the first (slot 0 runs on the thread
// that calls lgicp_submit).  They live as long as the context: a thread's first HIP calls and
// first allocations are slow, occasionally by milliseconds.
struct SlotThreads {
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::function<void(int)> job;  // called with the slot index
  uint64_t generation = 0;
  int wanted = 0;   // helpers that take part in the current job (slots 1..wanted)
  int pending = 0;
  bool stop = false;

  // owner: the context the helpers work for (where the buffers they outgrow are parked)
  void start(int n_helpers, struct ::lgicp_ctx* owner) {
    // a restarted set must not take the last job of the previous one for a new job
    const uint64_t born = generation;
    for (int k = 1; k <= n_helpers; ++k)
      threads.emplace_back([this, k, born, owner] {
        ParkScope park(owner);
        uint64_t seen = born;
        for (;;) {
          std::function<void(int)> fn;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv_job.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            if (k > wanted) continue;
            fn = job;
          }
          fn(k);
          {
            std::lock_guard<std::mutex> lk(mu);
            --pending;
          }
          cv_done.notify_all();
        }
      });
  }
  // runs fn(1..n) on the helpers; returns at once
  void post(int n, std::function<void(int)> fn) {
    {
      std::lock_guard<std::mutex> lk(mu);
      job = std::move(fn);
      wanted = pending = std::min<int>(n, (int)threads.size());
      ++generation;
    }
    cv_job.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return pending == 0; });
  }
  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_job.notify_all();
    for (std::thread& t : threads) t.join();
    threads.clear();
    stop = false;  // start() may be called again
    job = nullptr;
    wanted = pending = 0;
  }
};

struct lgicp_ctx {
  lgicp_config cfg;
  std::unordered_map<uint64_t, Keyframe> scans;
  std::unordered_map<uint64_t, hm::Pose> poses;
  std::string err;
  std::mutex err_mu;
  lgicp_stats stats;
  int nn_variant = 4;
  bool nn_exhaustive = false;  // reserved[2] = 1: the dense 1-NN kernel instead of the pruned one
  float nn_r0sq = 1.0f / 64.0f;  // squared radius of the pruned search's first round
  static constexpr int MAX_SLOTS = 8;
  BatchSlot slot[MAX_SLOTS];
  // parts of a submit in flight at once, one stream each (the GPU exposes 4 hardware queues
  // per process); reserved[7] overrides
  int n_slots = 4;
  int n_wide = 4;  // the first n_wide slots' streams were found to overlap: what one submit spreads over
  // Submits that may run side by side, each on its own share of the slots (lgicp_set_submit_lanes):
  // lane l owns slots [l * n_slots / n_lanes, (l + 1) * n_slots / n_lanes), the calling thread
  // runs the first of them and helpers[l] the others.
  static constexpr int MAX_LANES = lgicp_detail::LaneTable::MAX_LANES;
  lgicp_detail::LaneTable lanes;
  SlotThreads helpers[MAX_LANES];
  // keyframes and poses: submits read them (shared), registration and lgicp_align change them
  mutable std::shared_mutex data_mu;
  // Writers hold this while they wait for and hold data_mu, submits pass through it before they
  // take data_mu shared: a registration that has arrived goes before the submits that arrive
  // after it (the shared mutex alone prefers readers, and submits from several threads overlap
  // without a gap for as long as there is work).
  std::mutex gate_mu;
  std::mutex flush_mu;  // flush_registrations: one submit at a time
  mutable std::mutex stats_mu;
  hipStream_t stream = nullptr;  // = slot[0].stream: registration and the single-kernel entry points
  Registry reg;                  // keyframe registration
  // Working state of the feature-based initialisations, FEATURE_SETS copies of it: a FEATURES /
  // TEASERPP submit works with one set (set f runs on the streams of slots 4 f .. 4 f + 3), so two
  // such submits can run side by side.
  static constexpr int FEATURE_SETS = 2;
  // bit matrices and matches of the pairs of a TEASERPP group, reused from submit to submit
  std::vector<std::shared_ptr<lgicp_detail::TeaserJob>> teaser_slots[FEATURE_SETS];
  std::shared_ptr<lgicp_detail::MatcherDev> matcher[FEATURE_SETS];
  std::shared_ptr<lgicp_detail::FeatureScratch> feature_scratch;  // [FEATURE_SETS * FEATURE_LANES]
  std::shared_ptr<lgicp_detail::SacLane> sac_lanes;               // [FEATURE_SETS * FEATURE_LANES]
  std::mutex feat_mu;  // Keyframe::features: looked up and filled by one submit at a time
  // device and pinned buffers of the TEASER++ clique search: [0] the single-pair entry points (the
  // context's stream), [1 + f] feature set f; created on first use, released by lgicp_destroy
  lgicp_detail::CliqueBuffers* clique_bufs[FEATURE_SETS + 1] = {};
  // accumulated clouds by key (AccEntry); a submit looks its targets up here, builds the missing
  // ones on its own stream and keeps what it uses alive until its slot's next batch
  std::mutex acc_mu;
  std::unordered_map<uint64_t, std::shared_ptr<lgicp_detail::AccEntry>> acc_cache;
  size_t acc_bytes = 0;
  uint64_t acc_tick = 0;
  // outgrown buffers of this context, freed when one of its submits begins with nothing in flight
  std::mutex parked_mu;
  std::vector<void*> parked_dev, parked_pinned;
};

namespace lgicp_detail {

// Harris keypoints + FPFH descriptors of one device-resident cloud (the feature front end of
// GetSacInitialAlignment, IcpLoopComputation.cc:654-682)
struct CloudFeatures {
  std::vector<int> idx;       // keypoint indices, ascending
  std::vector<float> xyz;     // [k][3]
  std::vector<float> feat;    // [k][33]
  // the same on the device (one allocation, released with the last copy of this struct):
  // descriptors and keypoint coordinates stay where the matcher reads them
  std::shared_ptr<DevBlock> dev;
  const float* d_feat = nullptr;  // [k][33]
  const float* d_kxyz = nullptr;  // [k][3]
};
// Uploads `count` clouds and builds, on the device, their spatial index and (derive) what depends
// on a keyframe alone: covariances from the normals, records, ExtractNormals, NormalizePCloud
// constants.  Enqueued on the context's stream; flush_registrations() waits for it and picks the
// constants up (the keyframes must be in ctx->scans under their keys by then).
int register_keyframes(lgicp_ctx* ctx, const KfInput* in, size_t count, Keyframe* out, bool derive);
int flush_registrations(lgicp_ctx* ctx);
// `lane` selects the stream (the context's batch slot streams) and the scratch buffers: calls
// with different lanes may run at the same time from different host threads.  Lanes
// FEATURE_LANES * f .. FEATURE_LANES * f + 3 belong to feature set f.
int compute_cloud_features(lgicp_ctx* ctx, const float4* d_xyz, const float4* d_nrm, int n,
                           CloudFeatures* out, int lane = 0);
// (lane: stream and buffers, as for compute_cloud_features)
int sac_align_device(lgicp_ctx* ctx, const CloudFeatures& src, const CloudFeatures& tgt,
                     const float guess[16], uint32_t seed, float T_out[16], double* score, int lane = 0);
int teaser_align_device(lgicp_ctx* ctx, const CloudFeatures& src, const CloudFeatures& tgt,
                        const float guess[16], float T_out[16], int* used, int* n_matches,
                        int* n_clique);
// the same in two halves: matches and consistency graph pair by pair, then clique, rotation and
// translation for a group of pairs together
struct TeaserJob {
  int n_matches = 0;
  std::vector<double> src, dst;  // matched keypoints, [n_matches][3]
  // graph.n == 0 when there are too few matches to solve; graph.bits is filled only when the
  // clique search runs on the host, otherwise the bit matrix lives in d_bits alone
  lgicp::teaser::BitGraph graph;
  DevBuf<unsigned long long> d_bits;  // [graph.n][graph.words], reused from pair to pair
  DevBuf<double> d_sd;                // matched keypoints on the device: src [n][3], then dst [n][3]
  double* h_sd = nullptr;             // pinned landing area of d_sd
  size_t h_sd_cap = 0;
  TeaserJob() = default;
  TeaserJob(const TeaserJob&) = delete;
  TeaserJob& operator=(const TeaserJob&) = delete;
  ~TeaserJob() {
    d_bits.release(), d_sd.release();
    if (h_sd) (void)hipHostFree(h_sd);
  }
};
// working buffers of compute_cloud_features, one set per lane (several clouds are processed side
// by side on different streams), reused from cloud to cloud
struct FeatureScratch {
  DevBuf<float> resp, spfh;
  DevBuf<unsigned char> key, need;
  DevBuf<float4> kp, boxes;
  void release() {
    resp.release(), spfh.release(), key.release(), need.release(), kp.release(), boxes.release();
  }
};
constexpr int FEATURE_LANES = 4;
// buffers of the descriptor matcher, reused from pair to pair (work of successive pairs is
// ordered by the context's stream)
struct MatcherDev {
  DevBuf<float> part_d;
  DevBuf<int> part_i, nn;
  void release() { part_d.release(), part_i.release(), nn.release(); }
};
// buffers of one SAC-IA alignment, one set per lane (several pairs are aligned side by side on
// different streams), reused from pair to pair
struct SacLane {
  MatcherDev md;
  DevBuf<float4> sk, tk;
  DevBuf<float> hyp;
  DevBuf<double> ef;
  void release() { md.release(), sk.release(), tk.release(), hyp.release(), ef.release(); }
};
// teaser::greedy_clique on a bit matrix in device memory (teaser_graph.hip): core numbers and the
// windows of starts run on `st`; false when the device refuses
// (bufs: the working buffers, *bufs is created on first use and belongs to the caller, who
// releases it with clique_buffers_delete while its device is current or not -- it switches itself)
bool greedy_clique_device(CliqueBuffers** bufs, int device, hipStream_t st, const unsigned long long* d_bits,
                          int n, int words, std::vector<int>* clique);
void clique_buffers_delete(CliqueBuffers* b);
// the same for `count` graphs at once (every launch carries a step of all of them); entries with
// n == 0 are skipped and get an empty clique
struct CliqueGraph {
  const unsigned long long* d_bits;
  int n, words;
};
struct CliqueTimes {  // hipEvent durations and algorithmic bytes of the two kernels, accumulated
  double starts_ms = 0, starts_row_bytes = 0, core_ms = 0, core_matrix_bytes = 0;
  uint64_t starts_launches = 0, core_launches = 0;
};
// rerun_improving: ignore the member sets the windows kept and run every improving start once
// more on its own (the fallback for a start that found no free slot; a switch for tests)
bool greedy_clique_group(CliqueBuffers** bufs, int device, hipStream_t st, const CliqueGraph* graphs,
                         int count, std::vector<int>* cliques, CliqueTimes* times = nullptr,
                         bool rerun_improving = false);
int clique_device_max_vertices();
// first half: enqueues matcher and consistency graph of one pair on the context's stream and
// returns; teaser_prepare_collect (after the stream has been synchronised) moves the matched
// keypoints into job->src / job->dst
// (st: the stream the work is enqueued on, the context's own when null; one caller at a time, the
// matcher's buffers are shared)
// (fset: whose descriptor-matcher buffers the call uses)
int teaser_prepare_device(lgicp_ctx* ctx, const CloudFeatures& src, const CloudFeatures& tgt,
                          TeaserJob* job, hipStream_t st = nullptr, int fset = 0, int sub = 0);
void teaser_prepare_collect(TeaserJob* job);
// (st: the stream of the clique search, the context's own when null; fset: whose clique-search
// buffers the call uses, -1 = those of the single-pair entry points)
int teaser_finish_group(lgicp_ctx* ctx, TeaserJob* const* jobs, const float* const* guesses, int count,
                        float (*T_out)[16], int* used, int* n_clique, CliqueTimes* times = nullptr,
                        hipStream_t st = nullptr, int fset = -1);

// device address of the static scratch array behind lgicp_nn_search_static, looked up once per
// device; `device` must be the current device
int static_scratch_init(int device);

inline int fail(lgicp_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) {
    std::lock_guard<std::mutex> lock(ctx->err_mu);
    ctx->err = buf;
  } else
    g_create_error = buf;
  return code;
}

#define HIP_TRY(ctx, expr)                                                                    \
  do {                                                                                        \
    hipError_t e__ = (expr);                                                                  \
    if (e__ != hipSuccess)                                                                    \
      return fail(ctx, LGICP_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),    \
                  __FILE__, __LINE__);                                                        \
  } while (0)

}  // namespace lgicp_detail
