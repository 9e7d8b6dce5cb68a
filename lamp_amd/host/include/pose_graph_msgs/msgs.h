// Plain-struct images of the ROS messages on the boundary of the loop-computation stage
// (pose_graph_msgs/msg/{LoopCandidate,LoopCandidateArray,KeyedScan,PoseGraph,PoseGraphNode,
// PoseGraphEdge,LoopComputationStatus}.msg of the reference).  Field names and meaning follow
// the .msg files so that an adapter in a ROS build copies field by field.
#pragma once

#include <array>
#include <cstdint>
#include <string>
#include <vector>

namespace geometry_msgs {
struct Point {
  double x = 0, y = 0, z = 0;
};
struct Quaternion {
  double x = 0, y = 0, z = 0, w = 0;  // ROS default-constructs to all zeros
};
struct Pose {
  Point position;
  Quaternion orientation;
};
}  // namespace geometry_msgs

namespace pose_graph_msgs {

struct Header {
  double stamp = 0;  // seconds
  std::string frame_id;
};

// LoopCandidate.msg:1-18
struct LoopCandidate {
  Header header;
  uint64_t key_from = 0;
  uint64_t key_to = 0;
  geometry_msgs::Pose pose_from;
  geometry_msgs::Pose pose_to;
  int32_t type = 0;
  double value = 0;
};

// LoopCandidateArray.msg:1-4
struct LoopCandidateArray {
  Header header;
  int32_t originator = 0;
  std::vector<LoopCandidate> candidates;
};

// pcl::PointXYZINormal as decoded by pcl::fromROSMsg (IcpLoopComputation.cc:387-388)
struct PointXYZINormal {
  float x = 0, y = 0, z = 0;
  float intensity = 0;
  float normal_x = 0, normal_y = 0, normal_z = 0;
};
using PointCloud = std::vector<PointXYZINormal>;

// KeyedScan.msg:3-4
struct KeyedScan {
  uint64_t key = 0;
  PointCloud scan;
};

struct PoseGraphNode {
  uint64_t key = 0;
  std::string ID;
  geometry_msgs::Pose pose;
};

// PoseGraphEdge.msg:1-25
struct PoseGraphEdge {
  static constexpr int32_t ODOM = 1, LOOPCLOSE = 3;
  uint64_t key_from = 0;
  uint64_t key_to = 0;
  int32_t type = 0;
  geometry_msgs::Pose pose;
  std::array<double, 36> covariance{};  // row-major, rotX rotY rotZ x y z
  double range = 0;
  double range_error = 0;  // ICP fitness (IcpLoopComputation.cc:298)
};

struct PoseGraph {
  Header header;
  std::vector<PoseGraphNode> nodes;
  std::vector<PoseGraphEdge> edges;
};

// LoopComputationStatus.msg:1-6
struct LoopComputationStatus {
  static constexpr int32_t COMPLETED_ALL = 1;
  int32_t type = 0;
};

}  // namespace pose_graph_msgs
