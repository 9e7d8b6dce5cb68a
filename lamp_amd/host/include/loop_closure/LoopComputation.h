// Host-side mirror of lamp_loop_closure::LoopComputation
// (reference: loop_closure/include/loop_closure/LoopComputation.h:22-68,
// loop_closure/src/LoopComputation.cc).  Same members and semantics; the ROS node handle is
// replaced by a flat parameter map and the two publishers by callbacks, so the class can be
// driven exactly as the reference's tests drive it (callbacks called directly).
#pragma once

#include <functional>
#include <map>
#include <queue>
#include <string>
#include <vector>

#include "pose_graph_msgs/msgs.h"

namespace lamp_loop_closure {

// stand-in for the ROS parameter server: "<ns>/<key>" -> value (pu::Get semantics: a missing
// key makes LoadParameters fail, IcpLoopComputation.cc:58-167)
class ParameterServer {
 public:
  void Set(const std::string& key, double v) { values_[key] = v; }
  bool Get(const std::string& key, double& v) const {
    auto it = values_.find(key);
    if (it == values_.end()) return false;
    v = it->second;
    return true;
  }
  template <typename T>
  bool GetAs(const std::string& key, T& v) const {
    double d;
    if (!Get(key, d)) return false;
    v = static_cast<T>(d);
    return true;
  }
  // values of loop_closure/config/laser_parameters.yaml + precision_parameters.yaml:10-11
  static ParameterServer ReferenceDefaults();

 private:
  std::map<std::string, double> values_;
};

struct NodeHandle {
  std::string ns;  // "base" or "robot" (lamp_utils GetParamNamespace, PrefixHandling.h:108-121)
  const ParameterServer* params = nullptr;
};

struct Transform3 {  // geometry_utils::Transform3
  double translation[3] = {0, 0, 0};
  double rotation[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // row-major
};

class LoopComputation {
 public:
  LoopComputation();
  virtual ~LoopComputation();

  virtual bool Initialize(const NodeHandle& n) = 0;
  virtual bool LoadParameters(const NodeHandle& n);
  virtual bool CreatePublishers(const NodeHandle& n);
  virtual bool RegisterCallbacks(const NodeHandle& n);

  // Compute transform and populate output queue
  virtual void ComputeTransforms() = 0;

  void PublishLoopClosures();
  void InputCallback(const pose_graph_msgs::LoopCandidateArray& input_candidates);
  void PublishCompletedAllStatus();

  pose_graph_msgs::PoseGraphEdge CreateLoopClosureEdge(uint64_t key1, uint64_t key2,
                                                       const Transform3& delta,
                                                       const double covariance[36]) const;
  std::vector<pose_graph_msgs::PoseGraphEdge> GetCurrentOutputQueue();

  // where the two ROS publishers of the reference deliver (loop_closures,
  // loop_computation_status)
  std::function<void(const pose_graph_msgs::PoseGraph&)> loop_closure_pub_;
  std::function<void(const pose_graph_msgs::LoopComputationStatus&)> status_pub_;

 protected:
  std::vector<pose_graph_msgs::PoseGraphEdge> output_queue_;
  std::queue<pose_graph_msgs::LoopCandidate> input_queue_;
  double keyed_scans_max_delay_ = 0;
  std::string param_ns_;
};

}  // namespace lamp_loop_closure
