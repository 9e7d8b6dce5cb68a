// Host-side mirror of loop_closure/src/LoopComputation.cc of the reference.
#include "loop_closure/LoopComputation.h"

#include <cmath>

namespace lamp_loop_closure {

LoopComputation::LoopComputation() {}
LoopComputation::~LoopComputation() {}

bool LoopComputation::LoadParameters(const NodeHandle& n) {
  param_ns_ = n.ns;  // lamp_utils::GetParamNamespace(n.getNamespace())
  return n.params != nullptr;
}

bool LoopComputation::CreatePublishers(const NodeHandle&) { return true; }
bool LoopComputation::RegisterCallbacks(const NodeHandle&) { return true; }

void LoopComputation::PublishCompletedAllStatus() {
  pose_graph_msgs::LoopComputationStatus status;
  status.type = pose_graph_msgs::LoopComputationStatus::COMPLETED_ALL;
  if (status_pub_) status_pub_(status);
}

std::vector<pose_graph_msgs::PoseGraphEdge> LoopComputation::GetCurrentOutputQueue() {
  return output_queue_;
}

// LoopComputation.cc:47-57: publish the edges (if any), clear, then signal COMPLETED_ALL —
// the signal is what releases the next batch from the candidate queue
void LoopComputation::PublishLoopClosures() {
  if (output_queue_.empty()) {
    PublishCompletedAllStatus();
    return;
  }
  pose_graph_msgs::PoseGraph msg;
  msg.edges = output_queue_;
  if (loop_closure_pub_) loop_closure_pub_(msg);
  output_queue_.clear();
  PublishCompletedAllStatus();
}

void LoopComputation::InputCallback(const pose_graph_msgs::LoopCandidateArray& in) {
  for (const auto& c : in.candidates) input_queue_.push(c);
}

// LoopComputation.cc:67-87; gr::ToRosPose turns the rotation matrix into a quaternion
pose_graph_msgs::PoseGraphEdge LoopComputation::CreateLoopClosureEdge(
    uint64_t key1, uint64_t key2, const Transform3& delta, const double covariance[36]) const {
  pose_graph_msgs::PoseGraphEdge edge;
  edge.key_from = key1;
  edge.key_to = key2;
  edge.type = pose_graph_msgs::PoseGraphEdge::LOOPCLOSE;
  edge.pose.position.x = delta.translation[0];
  edge.pose.position.y = delta.translation[1];
  edge.pose.position.z = delta.translation[2];
  const double* R = delta.rotation;
  double q[4];  // w x y z
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    const double s = std::sqrt(tr + 1.0) * 2;
    q[0] = 0.25 * s, q[1] = (R[7] - R[5]) / s, q[2] = (R[2] - R[6]) / s, q[3] = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const double s = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
    q[0] = (R[7] - R[5]) / s, q[1] = 0.25 * s, q[2] = (R[1] + R[3]) / s, q[3] = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    const double s = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
    q[0] = (R[2] - R[6]) / s, q[1] = (R[1] + R[3]) / s, q[2] = 0.25 * s, q[3] = (R[5] + R[7]) / s;
  } else {
    const double s = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
    q[0] = (R[3] - R[1]) / s, q[1] = (R[2] + R[6]) / s, q[2] = (R[5] + R[7]) / s, q[3] = 0.25 * s;
  }
  edge.pose.orientation.w = q[0];
  edge.pose.orientation.x = q[1];
  edge.pose.orientation.y = q[2];
  edge.pose.orientation.z = q[3];
  for (size_t i = 0; i < 6; ++i)
    for (size_t j = 0; j < 6; ++j) edge.covariance[6 * i + j] = covariance[6 * i + j];
  return edge;
}

ParameterServer ParameterServer::ReferenceDefaults() {
  ParameterServer p;
  for (const char* ns : {"base", "robot"}) {
    const bool base = std::string(ns) == "base";
    const std::string s = std::string(ns) + "/";
    p.Set(s + "keyed_scans_max_delay", base ? 600.0 : 0.1);
    p.Set(s + "max_tolerable_fitness", base ? 0.9 : 0.4);
    p.Set(s + "distance_before_reclosing", 5);
    p.Set(s + "icp_lc/tf_epsilon", 0.0000000001);
    p.Set(s + "icp_lc/corr_dist", base ? 0.5 : 0.1);
    p.Set(s + "icp_lc/iterations", base ? 200 : 5);
    p.Set(s + "icp_lc/threads", base ? 8 : 4);
    // the robot block of the reference YAML lacks these three keys although LoadParameters
    // requires them (IcpLoopComputation.cc:81-87); they are given the base values here
    p.Set(s + "icp_lc/transform_thresholding", base ? 1 : 0);
    p.Set(s + "icp_lc/max_translation", 20);
    p.Set(s + "icp_lc/max_rotation", 50);
    p.Set(s + "sac_ia/iterations", base ? 500 : 10);
    p.Set(s + "sac_ia/num_prev_scans", 2);
    p.Set(s + "sac_ia/num_next_scans", 2);
    p.Set(s + "sac_ia/b_accumulate_source", base ? 0 : 1);
    p.Set(s + "sac_ia/features_radius", 2.0);
    p.Set(s + "sac_ia/fitness_score_threshold", base ? 32.0 : 1.2);
    p.Set(s + "harris3D/harris_threshold", 0.000000005);
    p.Set(s + "harris3D/harris_suppression", 1);
    p.Set(s + "harris3D/harris_radius", 0.1);
    p.Set(s + "harris3D/harris_refine", 0);
    p.Set(s + "harris3D/harris_response", 1);
    p.Set(s + "icp_initialization_method", 3);
    p.Set(s + "icp_covariance_calculation", 1);
    p.Set(s + "icp_thread_pool_thread_count", base ? 0.8 : 1);
  }
  p.Set("laser_lc_rot_sigma", 0.01);
  p.Set("laser_lc_trans_sigma", 0.01);
  p.Set("b_use_fixed_covariances", 0);
  return p;
}

}  // namespace lamp_loop_closure
