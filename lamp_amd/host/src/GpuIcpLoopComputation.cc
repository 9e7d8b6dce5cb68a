// GpuIcpLoopComputation — the reference's IcpLoopComputation driver logic
// (loop_closure/src/IcpLoopComputation.cc) on top of the lgicp C ABI.
#include "loop_closure/GpuIcpLoopComputation.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace lamp_loop_closure {

GpuIcpLoopComputation::GpuIcpLoopComputation() { lgicp_default_config(&cfg_, 1); }

GpuIcpLoopComputation::~GpuIcpLoopComputation() {
  for (lgicp_ctx* c : ctxs_) lgicp_destroy(c);
}

const char* GpuIcpLoopComputation::last_error() const { return lgicp_last_error(ctx_); }

// IcpLoopComputation::Initialize (IcpLoopComputation.cc:29-56); the thread pool of the
// reference is replaced by the device batch, so there is nothing to size here
bool GpuIcpLoopComputation::Initialize(const NodeHandle& n) {
  if (!LoadParameters(n)) {
    std::fprintf(stderr, "GpuIcpLoopComputation: Failed to load parameters.\n");
    return false;
  }
  if (!RegisterCallbacks(n)) return false;
  if (!CreatePublishers(n)) return false;
  for (lgicp_ctx* c : ctxs_) lgicp_destroy(c);
  ctxs_.clear();
  ctx_ = nullptr;
  keyed_scans_.clear();
  closed_keyes_.clear();
  host_scans_.clear();
  registered_.clear();
  std::vector<int> devices = devices_;
  if (devices.empty()) devices.push_back(cfg_.device);
  for (int d : devices) {
    lgicp_config cfg = cfg_;
    cfg.device = d;
    lgicp_ctx* c = nullptr;
    if (lgicp_create(&cfg, &c) != LGICP_OK) {
      std::fprintf(stderr, "GpuIcpLoopComputation: device %d: %s\n", d, lgicp_last_error(nullptr));
      for (lgicp_ctx* made : ctxs_) lgicp_destroy(made);
      ctxs_.clear();
      return false;
    }
    ctxs_.push_back(c);
  }
  ctx_ = ctxs_[0];
  registered_.resize(ctxs_.size());
  return true;
}

// IcpLoopComputation::LoadParameters (IcpLoopComputation.cc:58-167): every key is required
bool GpuIcpLoopComputation::LoadParameters(const NodeHandle& n) {
  if (!LoopComputation::LoadParameters(n)) return false;
  const ParameterServer& pu = *n.params;
  const std::string ns = param_ns_;
  const int device = cfg_.device;
  lgicp_default_config(&cfg_, ns == "base");
  cfg_.device = device;
  if (!pu.Get(ns + "/keyed_scans_max_delay", keyed_scans_max_delay_)) return false;
  if (!pu.Get(ns + "/max_tolerable_fitness", cfg_.max_tolerable_fitness)) return false;
  if (!pu.GetAs(ns + "/distance_before_reclosing", dist_before_reclosing_)) return false;
  if (!pu.Get(ns + "/icp_lc/tf_epsilon", cfg_.tf_epsilon)) return false;
  if (!pu.Get(ns + "/icp_lc/corr_dist", cfg_.corr_dist)) return false;
  if (!pu.GetAs(ns + "/icp_lc/iterations", cfg_.iterations)) return false;
  int icp_threads;  // OpenMP threads of the reference: no meaning on the device
  if (!pu.GetAs(ns + "/icp_lc/threads", icp_threads)) return false;
  if (!pu.GetAs(ns + "/icp_lc/transform_thresholding", cfg_.transform_thresholding)) return false;
  if (!pu.Get(ns + "/icp_lc/max_translation", cfg_.max_translation)) return false;
  if (!pu.Get(ns + "/icp_lc/max_rotation", cfg_.max_rotation_deg)) return false;
  if (!pu.GetAs(ns + "/sac_ia/iterations", cfg_.sac_iterations)) return false;
  if (!pu.GetAs(ns + "/sac_ia/num_prev_scans", cfg_.num_prev_scans)) return false;
  if (!pu.GetAs(ns + "/sac_ia/num_next_scans", cfg_.num_next_scans)) return false;
  if (!pu.GetAs(ns + "/sac_ia/b_accumulate_source", cfg_.accumulate_source)) return false;
  if (!pu.Get(ns + "/sac_ia/features_radius", cfg_.sac_features_radius)) return false;
  if (!pu.Get(ns + "/sac_ia/fitness_score_threshold", cfg_.sac_fitness_score_threshold))
    return false;
  if (!pu.Get(ns + "/harris3D/harris_threshold", cfg_.harris_threshold)) return false;
  if (!pu.GetAs(ns + "/harris3D/harris_suppression", cfg_.harris_suppression)) return false;
  if (!pu.Get(ns + "/harris3D/harris_radius", cfg_.harris_radius)) return false;
  if (!pu.GetAs(ns + "/harris3D/harris_refine", cfg_.harris_refine)) return false;
  if (!pu.GetAs(ns + "/harris3D/harris_response", cfg_.harris_response)) return false;
  if (!pu.GetAs(ns + "/icp_initialization_method", cfg_.init_method)) return false;
  if (!pu.GetAs(ns + "/icp_covariance_calculation", cfg_.covariance_method)) return false;
  if (!pu.Get("laser_lc_rot_sigma", cfg_.laser_lc_rot_sigma)) return false;
  if (!pu.Get("laser_lc_trans_sigma", cfg_.laser_lc_trans_sigma)) return false;
  if (!pu.GetAs("b_use_fixed_covariances", cfg_.use_fixed_covariances)) return false;
  double pool;  // host thread pool size of the reference: replaced by the device batch
  if (!pu.Get(ns + "/icp_thread_pool_thread_count", pool)) return false;
  // not in the reference (optional): the inner solver, and how many GPUs share the candidates
  pu.GetAs(ns + "/icp_lc/solver", cfg_.solver);
  int gpu_devices = 0;
  if (devices_.empty() && pu.GetAs(ns + "/gpu_devices", gpu_devices) && gpu_devices > 1)
    for (int d = 0; d < gpu_devices; ++d) devices_.push_back(d);
  return true;
}

bool GpuIcpLoopComputation::CreatePublishers(const NodeHandle& n) {
  return LoopComputation::CreatePublishers(n);
}
bool GpuIcpLoopComputation::RegisterCallbacks(const NodeHandle& n) {
  return LoopComputation::RegisterCallbacks(n);
}

// IcpLoopComputation::KeyedScanCallback (IcpLoopComputation.cc:377-392): first scan wins
void GpuIcpLoopComputation::KeyedScanCallback(const pose_graph_msgs::KeyedScan& m) {
  if (keyed_scans_.count(m.key)) return;
  if (ctxs_.size() > 1) {
    // which GPU needs the scan is known only when a candidate names it: keep it on the host
    host_scans_.emplace(m.key, m.scan);
    keyed_scans_.insert(m.key);
    return;
  }
  const pose_graph_msgs::PointXYZINormal* p = m.scan.data();
  const float* xyz = p ? &p->x : nullptr;
  const float* nrm = p ? &p->normal_x : nullptr;
  if (lgicp_add_keyframe(ctx_, m.key, xyz, nrm, m.scan.size(),
                         sizeof(pose_graph_msgs::PointXYZINormal)) >= 0)
    keyed_scans_.insert(m.key);
}

// IcpLoopComputation::KeyedPoseCallback (IcpLoopComputation.cc:394-418): first pose wins
void GpuIcpLoopComputation::KeyedPoseCallback(const pose_graph_msgs::PoseGraph& g) {
  for (const auto& node : g.nodes) {
    const double q[4] = {node.pose.orientation.w, node.pose.orientation.x,
                         node.pose.orientation.y, node.pose.orientation.z};
    const double t[3] = {node.pose.position.x, node.pose.position.y, node.pose.position.z};
    for (lgicp_ctx* c : ctxs_) lgicp_add_pose(c, node.key, q, t);  // 56 bytes: every device keeps all
  }
}

namespace {
void split(const pose_graph_msgs::PointCloud& c, std::vector<float>& xyz, std::vector<float>& nrm) {
  xyz.resize(c.size() * 3), nrm.resize(c.size() * 3);
  for (size_t i = 0; i < c.size(); ++i) {
    xyz[3 * i] = c[i].x, xyz[3 * i + 1] = c[i].y, xyz[3 * i + 2] = c[i].z;
    nrm[3 * i] = c[i].normal_x, nrm[3 * i + 1] = c[i].normal_y, nrm[3 * i + 2] = c[i].normal_z;
  }
}
}  // namespace

// IcpLoopComputation::GetTeaserInitialAlignment (IcpLoopComputation.cc:949-1081)
void GpuIcpLoopComputation::GetTeaserInitialAlignment(const pose_graph_msgs::PointCloud& source,
                                                      const pose_graph_msgs::PointCloud& target,
                                                      float tf_out[16]) {
  std::vector<float> sx, sn, tx, tn;
  split(source, sx, sn);
  split(target, tx, tn);
  float T[16];
  int32_t used = 0;
  if (lgicp_teaser_initial_alignment(ctx_, sx.data(), sn.data(), source.size(), tx.data(),
                                     tn.data(), target.size(), tf_out, T, &used, nullptr,
                                     nullptr) == LGICP_OK)
    std::memcpy(tf_out, T, sizeof(T));
}

// IcpLoopComputation::GetSacInitialAlignment (IcpLoopComputation.cc:650-705)
void GpuIcpLoopComputation::GetSacInitialAlignment(const pose_graph_msgs::PointCloud& source,
                                                   const pose_graph_msgs::PointCloud& target,
                                                   float tf_out[16],
                                                   double& sac_fitness_score) {
  std::vector<float> sx, sn, tx, tn;
  split(source, sx, sn);
  split(target, tx, tn);
  float T[16];
  if (lgicp_sac_initial_alignment(ctx_, sx.data(), sn.data(), source.size(), tx.data(), tn.data(),
                                  target.size(), tf_out, T, &sac_fitness_score) == LGICP_OK)
    std::memcpy(tf_out, T, sizeof(T));
}

// IcpLoopComputation::CheckReclosingDistance (IcpLoopComputation.cc:205-240): a pair is
// skipped when both of its keys lie within dist_before_reclosing_ of an already closed key.
// The reference dereferences std::prev(begin()) when the key precedes every closed key; the
// nearest closed key is taken here without that out-of-range read.
bool GpuIcpLoopComputation::CheckReclosingDistance(uint64_t key_from, uint64_t key_to) const {
  if (closed_keyes_.empty()) return true;
  auto nearest = [&](uint64_t key) {
    auto it = closed_keyes_.lower_bound(key);
    if (it == closed_keyes_.begin()) return *it;
    auto prev = std::prev(it);
    if (it == closed_keyes_.end() || key - *prev <= *it - key) return *prev;
    return *it;
  };
  const uint64_t closest_from = nearest(key_from), closest_to = nearest(key_to);
  const int df = std::abs(static_cast<int>(closest_from) - static_cast<int>(key_from));
  const int dt = std::abs(static_cast<int>(closest_to) - static_cast<int>(key_to));
  return !(df < static_cast<int>(dist_before_reclosing_) &&
           dt < static_cast<int>(dist_before_reclosing_));
}

static void fill_candidate(const pose_graph_msgs::LoopCandidate& c, lgicp_candidate* o) {
  o->key_from = c.key_from;
  o->key_to = c.key_to;
  const geometry_msgs::Pose* ps[2] = {&c.pose_from, &c.pose_to};
  double* qs[2] = {o->pose_from_q, o->pose_to_q};
  double* ts[2] = {o->pose_from_t, o->pose_to_t};
  for (int k = 0; k < 2; ++k) {
    qs[k][0] = ps[k]->orientation.w;
    qs[k][1] = ps[k]->orientation.x;
    qs[k][2] = ps[k]->orientation.y;
    qs[k][3] = ps[k]->orientation.z;
    ts[k][0] = ps[k]->position.x;
    ts[k][1] = ps[k]->position.y;
    ts[k][2] = ps[k]->position.z;
  }
}

// IcpLoopComputation::ComputeTransforms (IcpLoopComputation.cc:243-367).  Candidates whose
// scans have not arrived are re-queued while younger than keyed_scans_max_delay_; the rest
// go to the device as one batch.  The reclosing filter is order dependent (closed_keyes_ grows
// as edges are accepted), so it is applied on the host in queue order after the batch
// returns — the semantics of the reference's serial path (:274-297); its threaded path reads
// and writes closed_keyes_ concurrently without a lock.
void GpuIcpLoopComputation::ComputeTransforms() {
  const size_t n = input_queue_.size();
  std::vector<pose_graph_msgs::LoopCandidate> batch;
  for (size_t i = 0; i < n; i++) {
    auto candidate = input_queue_.front();
    input_queue_.pop();
    if (!keyed_scans_.count(candidate.key_from) || !keyed_scans_.count(candidate.key_to)) {
      if (now_() - candidate.header.stamp < keyed_scans_max_delay_) input_queue_.push(candidate);
      continue;
    }
    batch.push_back(candidate);
  }
  if (batch.empty()) return;
  std::vector<lgicp_candidate> cands(batch.size());
  std::vector<lgicp_result> results(batch.size());
  for (size_t i = 0; i < batch.size(); ++i) fill_candidate(batch[i], &cands[i]);
  if (ctxs_.size() > 1)
    SubmitSharded(cands, &results);
  else
    SubmitOne(ctx_, cands.data(), cands.size(), results.data());
  for (size_t i = 0; i < batch.size(); ++i) {
    const lgicp_result& r = results[i];
    if (!CheckReclosingDistance(r.key_from, r.key_to)) continue;
    if (r.status != LGICP_PAIR_OK) continue;
    Transform3 transform;
    std::memcpy(transform.translation, r.delta_t, sizeof(transform.translation));
    std::memcpy(transform.rotation, r.delta_R, sizeof(transform.rotation));
    pose_graph_msgs::PoseGraphEdge loop_closure =
        CreateLoopClosureEdge(r.key_from, r.key_to, transform, r.covariance);
    closed_keyes_.insert(r.key_from);
    closed_keyes_.insert(r.key_to);
    loop_closure.range_error = r.fitness;
    output_queue_.push_back(loop_closure);
  }
}

// One blocking batch on one context.  The reference fails per pair and never loses the rest of its
// queue (every `return false` of PerformAlignment drops one candidate): a call-level error of the
// batch is retried one candidate at a time, so that only the candidates the engine cannot process
// are dropped.
void GpuIcpLoopComputation::SubmitOne(lgicp_ctx* ctx, lgicp_candidate* cands, size_t n,
                                      lgicp_result* results) {
  if (n == 0 || lgicp_submit(ctx, cands, n, results) == LGICP_OK) return;
  std::fprintf(stderr, "GpuIcpLoopComputation: batch of %zu failed (%s); retrying per candidate\n", n,
               lgicp_last_error(ctx));
  for (size_t i = 0; i < n; ++i) {
    if (lgicp_submit(ctx, &cands[i], 1, &results[i]) == LGICP_OK) continue;
    std::fprintf(stderr, "GpuIcpLoopComputation: candidate %llu -> %llu dropped: %s\n",
                 (unsigned long long)cands[i].key_from, (unsigned long long)cands[i].key_to,
                 lgicp_last_error(ctx));
    results[i].key_from = cands[i].key_from;
    results[i].key_to = cands[i].key_to;
    results[i].status = LGICP_PAIR_NOT_CONVERGED;
  }
}

// The candidates of a tick cut over the contexts (the reference's pool of PerformAlignment workers,
// IcpLoopComputation.cc:327-365, across GPUs): partition by cost with affinity to the scans a
// device already holds, register what each device is missing -- the two scans of its pairs and the
// neighbours AccumulateScans will look for (IcpLoopComputation.cc:910-947: key +- i for the target,
// and for the source when b_accumulate_source is set) --, submit from one thread per context, and
// put the results back in queue order.
void GpuIcpLoopComputation::SubmitSharded(const std::vector<lgicp_candidate>& cands,
                                          std::vector<lgicp_result>* results) {
  const size_t world = ctxs_.size();
  std::vector<PairKeys> pairs(cands.size());
  for (size_t i = 0; i < cands.size(); ++i) pairs[i] = {cands[i].key_from, cands[i].key_to};
  auto cloud_size = [&](uint64_t key) {
    auto it = host_scans_.find(key);
    return it == host_scans_.end() ? size_t{0} : it->second.size();
  };
  std::vector<std::set<uint64_t>> held = registered_;
  const std::vector<std::vector<size_t>> share = PartitionPairs(pairs, cloud_size, (int)world, &held);
  std::vector<std::thread> workers;
  for (size_t d = 0; d < world; ++d) {
    if (share[d].empty()) continue;
    workers.emplace_back([&, d] {
      // scans device d is missing
      std::set<uint64_t> need;
      auto with_neighbours = [&](uint64_t key, bool accumulate) {
        need.insert(key);
        if (!accumulate) return;
        for (int i = 1; i <= cfg_.num_prev_scans; ++i)
          if (key >= (uint64_t)i) need.insert(key - i);
        for (int i = 1; i <= cfg_.num_next_scans; ++i) need.insert(key + i);
      };
      for (size_t i : share[d]) {
        with_neighbours(cands[i].key_to, true);
        with_neighbours(cands[i].key_from, cfg_.accumulate_source != 0);
      }
      std::vector<uint64_t> keys;
      std::vector<const float*> xyz, nrm;
      std::vector<size_t> counts;
      for (uint64_t k : need) {
        auto it = host_scans_.find(k);
        if (it == host_scans_.end() || registered_[d].count(k)) continue;
        const pose_graph_msgs::PointXYZINormal* p = it->second.data();
        keys.push_back(k);
        xyz.push_back(p ? &p->x : nullptr);
        nrm.push_back(p ? &p->normal_x : nullptr);
        counts.push_back(it->second.size());
      }
      if (!keys.empty() &&
          lgicp_add_keyframes(ctxs_[d], keys.size(), keys.data(), xyz.data(), nrm.data(), counts.data(),
                              sizeof(pose_graph_msgs::PointXYZINormal)) >= 0)
        registered_[d].insert(keys.begin(), keys.end());
      std::vector<lgicp_candidate> mine(share[d].size());
      std::vector<lgicp_result> out(share[d].size());
      for (size_t j = 0; j < share[d].size(); ++j) mine[j] = cands[share[d][j]];
      SubmitOne(ctxs_[d], mine.data(), mine.size(), out.data());
      for (size_t j = 0; j < share[d].size(); ++j) (*results)[share[d][j]] = out[j];
    });
  }
  for (std::thread& t : workers) t.join();
}

void GpuIcpLoopComputation::ProcessTimerCallback() {
  ComputeTransforms();
  if (loop_closure_pub_) PublishLoopClosures();
}

// IcpLoopComputation::PerformAlignment for a single pair (what the reference's tests call
// through their friend fixtures)
bool GpuIcpLoopComputation::PerformAlignment(uint64_t key1, uint64_t key2,
                                             const geometry_msgs::Pose& pose1,
                                             const geometry_msgs::Pose& pose2, Transform3* delta,
                                             double covariance[36], double* fitness_score) {
  if (delta == nullptr || covariance == nullptr) return false;
  pose_graph_msgs::LoopCandidate c;
  c.key_from = key1;
  c.key_to = key2;
  c.pose_from = pose1;
  c.pose_to = pose2;
  lgicp_candidate lc;
  lgicp_result r;
  fill_candidate(c, &lc);
  if (ctxs_.size() > 1) {
    std::vector<lgicp_candidate> one(1, lc);
    std::vector<lgicp_result> res(1);
    SubmitSharded(one, &res);
    r = res[0];
  } else if (lgicp_submit(ctx_, &lc, 1, &r) != LGICP_OK) {
    return false;
  }
  // delta is filled before the gates in the reference (:568-577): give the ICP transform
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) delta->rotation[a * 3 + b] = r.T_icp[a * 4 + b];
    delta->translation[a] = r.T_icp[a * 4 + 3];
  }
  if (r.status != LGICP_PAIR_OK) return false;
  std::memcpy(delta->translation, r.delta_t, sizeof(delta->translation));
  std::memcpy(delta->rotation, r.delta_R, sizeof(delta->rotation));
  std::memcpy(covariance, r.covariance, sizeof(double) * 36);
  if (fitness_score) *fitness_score = r.fitness;
  return true;
}

}  // namespace lamp_loop_closure
