"""Keyed-scan down-sampling: pcl::VoxelGrid / pcl::RandomSample / LampPcldFilter
(lamp_utils/src/LampPcldFilter.cc).  No reference test pins a filter output: parity unpinned;
the CPU restatement is checked through the properties PCL's algorithm guarantees, the device
path bit for bit against the restatement."""
import numpy as np
import pytest

from oracle import oracle as O
from lamp_amd.pcld_filter import LampPcldFilter, LampPcldFilterParams
from tests import fixtures as F


def _cloud(n, seed, span=6.0):
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-span, span, size=(n, 3)).astype(np.float32)
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return xyz, nrm


def test_voxel_grid_oracle_properties():
    xyz, nrm = _cloud(20000, 0, span=2.0)
    leaf = 0.5
    v, vn = O.voxel_grid(xyz, leaf, nrm)
    inv = np.float32(1.0) / np.float32(leaf)
    cell = np.floor(xyz * inv).astype(np.int64)
    cell -= np.floor(xyz.min(0) * inv).astype(np.int64)
    dims = cell.max(0) + 1
    lin = cell[:, 0] + dims[0] * (cell[:, 1] + dims[1] * cell[:, 2])
    uniq = np.unique(lin)
    assert len(v) == len(uniq) == 8 * 8 * 8                       # every cell is occupied
    # one output per cell, in ascending cell order, each inside its cell, equal to the mean
    out_cell = np.floor(v * inv).astype(np.int64) - np.floor(xyz.min(0) * inv).astype(np.int64)
    out_lin = out_cell[:, 0] + dims[0] * (out_cell[:, 1] + dims[1] * out_cell[:, 2])
    assert np.array_equal(out_lin, uniq)
    for k in (0, 17, 300, 511):
        sel = lin == uniq[k]
        assert np.allclose(v[k], xyz[sel].astype(np.float64).mean(0), atol=2e-6)
        s = nrm[sel].astype(np.float64).sum(0)
        assert np.allclose(vn[k], s / np.linalg.norm(s), atol=1e-5)
    assert np.allclose(np.linalg.norm(vn, axis=1), 1, atol=1e-6)


def test_voxel_grid_oracle_edge_cases():
    xyz, nrm = _cloud(500, 1)
    # without normals, with non-finite points, empty input
    v = O.voxel_grid(xyz, 1.0)
    bad = xyz.copy()
    bad[::7, 1] = np.nan
    bad[3, 0] = np.inf
    keep = np.isfinite(bad).all(1)
    assert np.array_equal(O.voxel_grid(bad, 1.0), O.voxel_grid(bad[keep], 1.0))
    assert len(O.voxel_grid(np.zeros((0, 3), np.float32), 1.0)) == 0
    assert len(O.voxel_grid(np.full((4, 3), np.nan, np.float32), 1.0)) == 0
    # all points in one cell -> their mean; a leaf far larger than the cloud does the same
    one = O.voxel_grid(xyz * 1e-3 + 0.5, 1.0)
    assert one.shape == (1, 3)
    # a leaf so small that the grid needs more than INT_MAX cells hands the input back
    same, same_n = O.voxel_grid(xyz, 1e-3, nrm)
    assert np.array_equal(same, xyz) and np.array_equal(same_n, nrm)
    # zero normals stay zero
    _, zn = O.voxel_grid(xyz, 1.0, np.zeros_like(nrm))
    assert not zn.any()
    assert len(v) <= len(xyz)


def test_random_sample_oracle():
    for n, keep in ((1000, 100), (1000, 999), (5, 10), (5, 0), (0, 3)):
        idx = O.random_sample_indices(n, keep, 7)
        assert len(idx) == min(n, keep)
        assert np.all(np.diff(idx) > 0) and (len(idx) == 0 or (idx[0] >= 0 and idx[-1] < n))
    a, b = O.random_sample_indices(1000, 100, 7), O.random_sample_indices(1000, 100, 8)
    assert np.array_equal(a, O.random_sample_indices(1000, 100, 7)) and not np.array_equal(a, b)


def test_filter_state_machine_oracle():
    """LampPcldFilter.cc:36-72: the leaf size starts midway between the limits, is multiplied by
    (points out / target) after every scan and clamped; scans below the target pass through."""
    p = LampPcldFilterParams(random_filter=True, decimate_percentage=0.5, adaptive_grid_target=400,
                             adaptive_min_grid=0.1, adaptive_max_grid=2.0)
    f = O.LampPcldFilterOracle(p, seed=3)
    assert f.leaf == 1.05
    xyz, nrm = _cloud(6000, 2)
    leafs, sizes = [], []
    for _ in range(6):
        leaf = f.leaf
        x, n = f.filter(xyz, nrm)
        grid = O.voxel_grid(xyz, leaf)
        assert len(x) == int(0.5 * len(grid)) and len(n) == len(x)
        assert f.leaf == min(2.0, max(0.1, leaf * (len(grid) / 400.0)))
        leafs.append(f.leaf), sizes.append(len(grid))
    assert abs(sizes[-1] - 400) < abs(sizes[0] - 400)              # steered towards the target
    small = _cloud(300, 3)
    leaf = f.leaf
    x, n = f.adaptive_grid_filter(*small)
    assert np.array_equal(x, small[0]) and f.leaf == leaf
    # observability term: the reference compares every scan with itself, so it changes nothing
    # unless the observability is exactly zero (0/0 -> NaN -> the lower clamp)
    q = LampPcldFilterParams(observability_check=True, adaptive_grid_target=400,
                             adaptive_min_grid=0.1, adaptive_max_grid=2.0)
    g, h = O.LampPcldFilterOracle(q), O.LampPcldFilterOracle(p)
    g.adaptive_grid_filter(xyz, nrm), h.adaptive_grid_filter(xyz, nrm)
    assert g.leaf == h.leaf
    plane, pn = F.generate_plane()                                 # one normal direction: min eigenvalue 0
    flat = np.tile(plane, (5, 1)) + np.random.default_rng(0).normal(scale=1e-3, size=(500, 3)).astype(np.float32)
    g.adaptive_grid_filter(flat, np.tile(pn, (5, 1)))
    assert g.leaf == 0.1


class _CheckerEngine:
    """Stands in for the engine context so that the host-side state machine of
    lamp_amd.pcld_filter can be exercised without a GPU: same three calls, answered by the CPU
    restatement."""

    def voxel_grid(self, xyz, leaf, normals=None):
        return O.voxel_grid(xyz, leaf, normals)

    def random_sample(self, xyz, n_keep, seed=0, normals=None):
        idx = O.random_sample_indices(len(xyz), n_keep, seed)
        return (xyz[idx], normals[idx]) if normals is not None else xyz[idx]

    def icp_observability(self, xyz, normals):
        return O.icp_observability(xyz, normals)


@pytest.mark.parametrize("observability", [False, True])
def test_product_filter_host_logic(observability):
    """lamp_amd.pcld_filter.LampPcldFilter (the host mirror the product ships) drives its engine
    exactly as the restated reference class drives PCL: same outputs, same leaf size after
    every scan, pass-through below the target, NaN observability falling to the lower clamp."""
    p = LampPcldFilterParams(random_filter=True, decimate_percentage=0.7, adaptive_grid_target=800,
                             adaptive_min_grid=0.1, adaptive_max_grid=1.5,
                             observability_check=observability)
    dev, ref = LampPcldFilter(_CheckerEngine(), p, seed=5), O.LampPcldFilterOracle(p, seed=5)
    assert dev.grid_leaf_size == ref.leaf == 0.8
    for k in range(4):
        scan, sn = F.random_room_scan(9000, 30 + k)
        x, n = dev.filter(scan, sn)
        rx, rn = ref.filter(scan, sn)
        assert np.array_equal(x, rx) and np.array_equal(n, rn) and dev.grid_leaf_size == ref.leaf
    few, fn = _cloud(100, 9)
    x, n = dev.filter(few, fn)
    assert len(x) == int(0.3 * 100) and dev.grid_leaf_size == ref.leaf
    if observability:
        plane, pn = F.generate_plane()
        flat = np.tile(plane, (9, 1)) + np.random.default_rng(1).normal(scale=1e-3, size=(900, 3)).astype(np.float32)
        dev.adaptive_grid_filter(flat, np.tile(pn, (9, 1)))
        assert dev.grid_leaf_size == 0.1


# ----------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("n,leaf,seed", [(20000, 0.5, 0), (65536, 0.25, 1), (3000, 3.0, 2), (1, 1.0, 3)])
def test_voxel_grid_device_matches_oracle(n, leaf, seed):
    from lamp_amd import capi
    xyz, nrm = _cloud(n, seed)
    xyz[::11] = xyz[::11].round(1)                                  # points on cell faces
    c = capi.Context()
    try:
        v, vn = c.voxel_grid(xyz, leaf, nrm)
        rv, rn = O.voxel_grid(xyz, leaf, nrm)
        assert np.array_equal(v, rv) and np.array_equal(vn, rn)
        assert np.array_equal(c.voxel_grid(xyz, leaf), rv)
    finally:
        c.close()


@pytest.mark.gpu
def test_voxel_grid_device_edge_cases():
    from lamp_amd import capi
    xyz, nrm = _cloud(5000, 4)
    c = capi.Context()
    try:
        bad = xyz.copy()
        bad[::7, 1] = np.nan
        bad[3, 0] = np.inf
        assert np.array_equal(c.voxel_grid(bad, 1.0), O.voxel_grid(bad, 1.0))
        assert len(c.voxel_grid(np.zeros((0, 3), np.float32), 1.0)) == 0
        assert len(c.voxel_grid(np.full((4, 3), np.nan, np.float32), 1.0)) == 0
        same, same_n = c.voxel_grid(xyz, 1e-3, nrm)
        assert np.array_equal(same, xyz) and np.array_equal(same_n, nrm)
        _, zn = c.voxel_grid(xyz, 1.0, np.zeros_like(nrm))
        assert not zn.any()
        with pytest.raises(capi.LgicpError):
            c.voxel_grid(xyz, 0.0)
        # a real scan: VLP-16-like rings, clustered cells of very different sizes
        scan, sn = F.random_room_scan(30000, 5)
        v, vn = c.voxel_grid(scan, 0.4, sn)
        rv, rn = O.voxel_grid(scan, 0.4, sn)
        assert np.array_equal(v, rv) and np.array_equal(vn, rn)
    finally:
        c.close()


@pytest.mark.gpu
def test_random_sample_device_matches_oracle():
    from lamp_amd import capi
    xyz, nrm = _cloud(50000, 6)
    c = capi.Context()
    try:
        for keep, seed in ((5000, 0), (49999, 1), (1, 2), (0, 3), (60000, 4)):
            idx = O.random_sample_indices(len(xyz), keep, seed)
            x, n = c.random_sample(xyz, keep, seed, nrm)
            assert np.array_equal(x, xyz[idx]) and np.array_equal(n, nrm[idx])
        assert np.array_equal(c.random_sample(xyz, 777, 9), xyz[O.random_sample_indices(len(xyz), 777, 9)])
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("observability", [False, True])
def test_lamp_pcld_filter_device_matches_oracle(observability):
    """A sequence of keyed scans through LampPcldFilter::Filter: same clouds and the same leaf
    size after every scan as the CPU restatement."""
    from lamp_amd import capi
    p = LampPcldFilterParams(random_filter=True, decimate_percentage=0.6, adaptive_grid_target=1500,
                             adaptive_min_grid=0.1, adaptive_max_grid=1.0,
                             observability_check=observability)
    c = capi.Context()
    try:
        dev, ref = LampPcldFilter(c, p, seed=11), O.LampPcldFilterOracle(p, seed=11)
        for k in range(5):
            scan, sn = F.random_room_scan(20000, 20 + k)
            x, n = dev.filter(scan, sn)
            rx, rn = ref.filter(scan, sn)
            assert np.array_equal(x, rx) and np.array_equal(n, rn)
            assert dev.grid_leaf_size == ref.leaf
        assert dev.grid_leaf_size != 0.55                          # the controller moved
    finally:
        c.close()
