// LampPcldFilter — the reference's keyed-scan filter interface
// (lamp_utils/include/lamp_utils/LampPcldFilter.h:13-46) with the per-point work on the MI355X
// engine: same parameter struct, same Filter() entry point, same leaf-size controller; the
// voxel grid, the random sample and the observability run behind the C ABI of lamp_gicp.h.
#pragma once

#include <cstdint>
#include <memory>

#include "lamp_gicp.h"
#include "pose_graph_msgs/msgs.h"

struct LampPcldFilterParams {
  bool random_filter = true;
  // Percentage of points to discard. Must be between 0.0 and 1.0;
  double decimate_percentage = 0.9;
  bool adaptive_grid_filter = false;
  int adaptive_grid_target = 1000;
  double adaptive_max_grid = 1.0;
  double adaptive_min_grid = 0.1;
  bool observability_check = false;
};

class LampPcldFilter {
 public:
  using PointCloud = pose_graph_msgs::PointCloud;

  // ctx: an engine context that outlives the filter.  seed: the reference's PCL sampler seeds
  // itself from the clock; here scan number i of this filter is drawn with seed + i.
  LampPcldFilter(lgicp_ctx* ctx, const LampPcldFilterParams& params, uint32_t seed = 0);

  // false when the engine reports an error (text in lgicp_last_error); new_cloud is then
  // left untouched
  bool Filter(const PointCloud& original_cloud, PointCloud* new_cloud);

  double grid_leaf_size() const { return grid_leaf_size_; }

 private:
  bool AdaptiveGridFilter(double target_pt_size, double min_leaf_size, double max_leaf_size,
                          const PointCloud& original_cloud, PointCloud* new_cloud);

  lgicp_ctx* ctx_;
  LampPcldFilterParams params_;
  double grid_leaf_size_;
  double prev_observability_ = 0.0;
  bool processed_first_cloud_ = false;  // never set by the reference either (LampPcldFilter.cc:60)
  uint32_t seed_;
  uint32_t n_filtered_ = 0;
};
