// GpuIcpLoopComputation — sibling of lamp_loop_closure::IcpLoopComputation
// (reference: loop_closure/include/loop_closure/IcpLoopComputation.h:24-159) whose
// ComputeTransforms() hands the whole candidate queue to the MI355X engine through the C ABI of
// include/lamp_gicp.h instead of running PCL GICP on a host thread pool.
#pragma once

#include <map>
#include <set>
#include <unordered_map>
#include <vector>

#include "lamp_gicp.h"
#include "loop_closure/LoopComputation.h"
#include "loop_closure/PairSharding.h"

namespace lamp_loop_closure {

class GpuIcpLoopComputation : public LoopComputation {
 public:
  GpuIcpLoopComputation();
  ~GpuIcpLoopComputation() override;

  bool Initialize(const NodeHandle& n) override;
  bool LoadParameters(const NodeHandle& n) override;
  bool CreatePublishers(const NodeHandle& n) override;
  bool RegisterCallbacks(const NodeHandle& n) override;

  // Compute transform and populate output queue
  void ComputeTransforms() override;

  void KeyedScanCallback(const pose_graph_msgs::KeyedScan& scan_msg);
  void KeyedPoseCallback(const pose_graph_msgs::PoseGraph& graph_msg);
  void ProcessTimerCallback();

  bool PerformAlignment(uint64_t key1, uint64_t key2, const geometry_msgs::Pose& pose1,
                        const geometry_msgs::Pose& pose2, Transform3* delta, double covariance[36],
                        double* fitness_score);

  // IcpLoopComputation::GetSacInitialAlignment (IcpLoopComputation.h:110-113): *tf_out is the
  // guess on entry (row-major 4x4) and the SAC-IA transform on return
  void GetSacInitialAlignment(const pose_graph_msgs::PointCloud& source,
                              const pose_graph_msgs::PointCloud& target, float tf_out[16],
                              double& sac_fitness_score);

  // IcpLoopComputation::GetTeaserInitialAlignment (IcpLoopComputation.h:68-70)
  void GetTeaserInitialAlignment(const pose_graph_msgs::PointCloud& source,
                                 const pose_graph_msgs::PointCloud& target, float tf_out[16]);

  bool CheckReclosingDistance(uint64_t key_from, uint64_t key_to) const;

  // Several GPUs behind one node (the reference node is one process whose pool spreads the pairs of
  // a tick, loop_computation_node.cc:12-25, IcpLoopComputation.cc:327-365): one engine context per
  // entry of `devices` (HIP device ordinals; an ordinal may repeat, which gives two contexts on one
  // GPU).  Call before Initialize; the parameter "<ns>/gpu_devices" (a count: devices 0..count-1)
  // does the same from the parameter server.  With more than one context the keyed scans are kept
  // on the host, as the reference keeps them (IcpLoopComputation.h:97), every ComputeTransforms
  // cuts its candidates over the contexts (PartitionPairs), registers on each device only the
  // scans its pairs (and their accumulated neighbours) need and has not got yet, submits from one
  // thread per context and merges the results in queue order before the reclosing filter.
  void SetDevices(const std::vector<int>& devices) { devices_ = devices; }
  size_t device_count() const { return ctxs_.size(); }
  // keyed scans resident on context `d` (multi-context operation)
  size_t resident_scans(size_t d) const { return d < registered_.size() ? registered_[d].size() : 0; }

  // seconds; tests replace it (the reference calls ros::Time::now())
  std::function<double()> now_ = [] { return 0.0; };
  const lgicp_config& config() const { return cfg_; }
  const char* last_error() const;

 protected:
  void SubmitSharded(const std::vector<lgicp_candidate>& cands, std::vector<lgicp_result>* results);
  void SubmitOne(lgicp_ctx* ctx, lgicp_candidate* cands, size_t n, lgicp_result* results);

  lgicp_config cfg_;
  lgicp_ctx* ctx_ = nullptr;             // = ctxs_[0]: the single-pair entry points
  std::vector<lgicp_ctx*> ctxs_;         // one per entry of devices_
  std::vector<int> devices_;             // empty: the one device of cfg_
  std::map<uint64_t, pose_graph_msgs::PointCloud> host_scans_;  // more than one context only
  std::vector<std::set<uint64_t>> registered_;                  // per context
  unsigned int dist_before_reclosing_ = 0;
  std::set<uint64_t> closed_keyes_;
  std::set<uint64_t> keyed_scans_;  // keys with a scan on the device
};

}  // namespace lamp_loop_closure
