// GpuIcpLoopComputation — sibling of lamp_loop_closure::IcpLoopComputation
// (reference: loop_closure/include/loop_closure/IcpLoopComputation.h:24-159) whose
// ComputeTransforms() hands the whole candidate queue to the MI355X engine through the C ABI of
// include/lamp_gicp.h instead of running PCL GICP on a host thread pool.
#pragma once

#include <set>
#include <unordered_map>

#include "lamp_gicp.h"
#include "loop_closure/LoopComputation.h"

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

  // seconds; tests replace it (the reference calls ros::Time::now())
  std::function<double()> now_ = [] { return 0.0; };
  const lgicp_config& config() const { return cfg_; }
  const char* last_error() const;

 protected:
  lgicp_config cfg_;
  lgicp_ctx* ctx_ = nullptr;
  unsigned int dist_before_reclosing_ = 0;
  std::set<uint64_t> closed_keyes_;
  std::set<uint64_t> keyed_scans_;  // keys with a scan on the device
};

}  // namespace lamp_loop_closure
