"""Keyed-scan down-sampling on the GPU engine, behind the reference's LampPcldFilter interface
(lamp_utils/include/lamp_utils/LampPcldFilter.h:13-46, lamp_utils/src/LampPcldFilter.cc:12-72;
used by the robot for every keyed scan, lamp/src/LampRobot.cc).

The class keeps the reference's state machine on the host — the leaf size that is steered from
scan to scan — and sends the per-point work (voxel grid, random sample, observability) to the
device through the C ABI.  There is no CPU fallback."""
from dataclasses import dataclass

import numpy as np


@dataclass
class LampPcldFilterParams:
    """Same names and defaults as the reference's struct (LampPcldFilter.h:13-26)."""
    random_filter: bool = True
    decimate_percentage: float = 0.9
    adaptive_grid_filter: bool = False
    adaptive_grid_target: int = 1000
    adaptive_max_grid: float = 1.0
    adaptive_min_grid: float = 0.1
    observability_check: bool = False


class LampPcldFilter:
    def __init__(self, ctx, params=None, seed=0):
        self.ctx = ctx
        self.params = params or LampPcldFilterParams()
        # LampPcldFilter.cc:14
        self.grid_leaf_size = (self.params.adaptive_max_grid + self.params.adaptive_min_grid) / 2.0
        self.prev_observability = 0.0
        # the reference never sets this flag after the first cloud (LampPcldFilter.cc:60-61), so
        # its observability term compares every scan with itself; kept as it is
        self.processed_first_cloud = False
        self.seed = seed
        self.n_filtered = 0

    def filter(self, xyz, normals):
        """LampPcldFilter::Filter (LampPcldFilter.cc:17-34).  Returns (xyz, normals)."""
        xyz, normals = self.adaptive_grid_filter(xyz, normals)
        if self.params.random_filter:
            n_points = int((1.0 - self.params.decimate_percentage) * len(xyz))
            # PCL seeds its draw from the clock; here: base seed + number of scans filtered so far
            xyz, normals = self.ctx.random_sample(xyz, n_points, self.seed + self.n_filtered, normals)
        self.n_filtered += 1
        return xyz, normals

    def adaptive_grid_filter(self, xyz, normals):
        """LampPcldFilter::AdaptiveGridFilter (LampPcldFilter.cc:36-72): one voxel-grid pass with
        the current leaf size, then the leaf size is steered towards the target point count."""
        p = self.params
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        normals = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        if len(xyz) < p.adaptive_grid_target:
            return xyz, normals
        xyz, normals = self.ctx.voxel_grid(xyz, self.grid_leaf_size, normals)
        size_factor = len(xyz) / float(p.adaptive_grid_target)
        obs_factor = 0.0
        if p.observability_check:
            ev = self.ctx.icp_observability(xyz, normals)
            observability = float(np.min(ev)) / float(len(xyz))
            if not self.processed_first_cloud:
                self.prev_observability = observability
            with np.errstate(divide="ignore", invalid="ignore"):
                obs_factor = float((np.float64(self.prev_observability) - observability) /
                                   np.float64(self.prev_observability))
            self.prev_observability = observability
        self.grid_leaf_size = _clamp_like_std(
            self.grid_leaf_size * (size_factor - abs(size_factor - 1) * obs_factor),
            p.adaptive_min_grid, p.adaptive_max_grid)
        return xyz, normals


def _clamp_like_std(v, lo, hi):
    """std::min(hi, std::max(lo, v)) including what it does with NaN (a NaN v gives lo)."""
    m = v if lo < v else lo      # std::max(lo, v)
    return m if m < hi else hi   # std::min(hi, m)
