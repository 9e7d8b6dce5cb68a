// LampPcldFilter on the GPU engine (reference: lamp_utils/src/LampPcldFilter.cc:12-72).
#include "lamp_utils/LampPcldFilter.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace {

// pose_graph_msgs::PointCloud <-> the packed arrays of the C ABI.  The engine averages
// coordinates and normals; intensity does not take part in loop closure and is dropped.
void unpack(const LampPcldFilter::PointCloud& c, std::vector<float>* xyz, std::vector<float>* nrm) {
  xyz->resize(3 * c.size());
  nrm->resize(3 * c.size());
  for (size_t i = 0; i < c.size(); ++i) {
    (*xyz)[3 * i] = c[i].x, (*xyz)[3 * i + 1] = c[i].y, (*xyz)[3 * i + 2] = c[i].z;
    (*nrm)[3 * i] = c[i].normal_x, (*nrm)[3 * i + 1] = c[i].normal_y, (*nrm)[3 * i + 2] = c[i].normal_z;
  }
}

void pack(const std::vector<float>& xyz, const std::vector<float>& nrm, size_t n,
          LampPcldFilter::PointCloud* c) {
  c->assign(n, pose_graph_msgs::PointXYZINormal());
  for (size_t i = 0; i < n; ++i) {
    (*c)[i].x = xyz[3 * i], (*c)[i].y = xyz[3 * i + 1], (*c)[i].z = xyz[3 * i + 2];
    (*c)[i].normal_x = nrm[3 * i], (*c)[i].normal_y = nrm[3 * i + 1], (*c)[i].normal_z = nrm[3 * i + 2];
  }
}

}  // namespace

LampPcldFilter::LampPcldFilter(lgicp_ctx* ctx, const LampPcldFilterParams& params, uint32_t seed)
    : ctx_(ctx), params_(params), seed_(seed) {
  grid_leaf_size_ = (params.adaptive_max_grid + params.adaptive_min_grid) / 2.0;
}

bool LampPcldFilter::Filter(const PointCloud& original_cloud, PointCloud* new_cloud) {
  PointCloud out;
  if (!AdaptiveGridFilter(params_.adaptive_grid_target, params_.adaptive_min_grid,
                          params_.adaptive_max_grid, original_cloud, &out))
    return false;
  // Apply random downsampling to the keyed scan
  if (params_.random_filter) {
    const int n_points = static_cast<int>((1.0 - params_.decimate_percentage) * out.size());
    std::vector<float> xyz, nrm;
    unpack(out, &xyz, &nrm);
    std::vector<float> oxyz(xyz.size()), onrm(nrm.size());
    size_t m = 0;
    if (lgicp_random_sample(ctx_, xyz.data(), nrm.data(), out.size(), (size_t)std::max(n_points, 0),
                            seed_ + n_filtered_, oxyz.data(), onrm.data(), &m) != LGICP_OK)
      return false;
    pack(oxyz, onrm, m, &out);
  }
  ++n_filtered_;
  *new_cloud = std::move(out);
  return true;
}

bool LampPcldFilter::AdaptiveGridFilter(double target_pt_size, double min_leaf_size,
                                        double max_leaf_size, const PointCloud& original_cloud,
                                        PointCloud* new_cloud) {
  if (original_cloud.size() < target_pt_size) {
    *new_cloud = original_cloud;
    return true;
  }
  std::vector<float> xyz, nrm;
  unpack(original_cloud, &xyz, &nrm);
  std::vector<float> oxyz(xyz.size()), onrm(nrm.size());
  size_t m = 0;
  if (lgicp_voxel_grid(ctx_, xyz.data(), nrm.data(), original_cloud.size(), grid_leaf_size_,
                       oxyz.data(), onrm.data(), &m) != LGICP_OK)
    return false;
  pack(oxyz, onrm, m, new_cloud);

  const double size_factor = static_cast<double>(m) / target_pt_size;
  double obs_factor = 0.0;
  if (params_.observability_check) {
    double ev[3];
    if (lgicp_icp_observability(ctx_, oxyz.data(), onrm.data(), m, ev) != LGICP_OK) return false;
    const double observability = std::min(ev[0], std::min(ev[1], ev[2])) / static_cast<double>(m);
    if (!processed_first_cloud_) prev_observability_ = observability;
    obs_factor = (prev_observability_ - observability) / prev_observability_;
    prev_observability_ = observability;
  }
  grid_leaf_size_ = std::min(
      max_leaf_size,
      std::max(min_leaf_size, grid_leaf_size_ * (size_factor - std::fabs(size_factor - 1) * obs_factor)));
  return true;
}
