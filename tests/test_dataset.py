"""The reference's evaluation data-set format (loop_closure/src/TestUtils.cc) and summary."""
import os

import numpy as np
import pytest

from lamp_amd import dataset as D


def _small_data():
    d = D.TestData()
    rng = np.random.default_rng(0)
    for k in range(3):
        xyz = rng.normal(size=(50 + k, 3)).astype(np.float32)
        nrm = rng.normal(size=(50 + k, 3)).astype(np.float32)
        d.scans.append({"xyz": xyz, "normals": nrm, "intensity": np.arange(50 + k, dtype=np.float32)})
        d.gt_poses.append(((1.0, 0.0, 0.0, 0.0), (float(k), 0.0, 0.0)))
        d.odom_poses.append(((0.9238795325112867, 0.0, 0.0, 0.3826834323650898), (k + 0.1, 0.2, 0.0)))
        d.labels.append("a" if k < 2 else "b")
    d.real_candidates = [(0, 1), (2, 0)]
    d.fake_candidates = [(1, 2)]
    return d


def test_data_set_round_trip(tmp_path):
    d = _small_data()
    D.write_test_data(str(tmp_path), d)
    assert sorted(os.listdir(tmp_path)) == ["0.pcd", "1.pcd", "2.pcd", "candidates.csv",
                                            "false_candidates.csv", "gt_keyed_poses.csv",
                                            "keyed_poses.csv"]
    e = D.load_test_data(str(tmp_path))
    assert e.real_candidates == d.real_candidates and e.fake_candidates == d.fake_candidates
    assert e.labels == d.labels
    assert np.allclose(np.array([p[0] + p[1] for p in e.odom_poses]),
                       np.array([p[0] + p[1] for p in d.odom_poses]), atol=0, rtol=0)
    for a, b in zip(d.scans, e.scans):
        assert np.array_equal(a["xyz"], b["xyz"]) and np.array_equal(a["normals"], b["normals"])
        assert np.array_equal(a["intensity"], b["intensity"])


def test_pcd_reader_handles_pcl_padding_and_ascii(tmp_path):
    """pcl::io::savePCDFileBinary writes PointXYZINormal with padding fields (`_`, SIZE 1,
    COUNT n) between the members; ascii files list the same fields as text."""
    n = 4
    dt = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("p0", "u1", (4,)),
                   ("normal_x", "<f4"), ("normal_y", "<f4"), ("normal_z", "<f4"), ("p1", "u1", (4,)),
                   ("intensity", "<f4"), ("curvature", "<f4"), ("p2", "u1", (8,))])
    rec = np.zeros(n, dt)
    rec["x"], rec["normal_z"], rec["intensity"] = np.arange(n), 1.0, 7.0
    p = tmp_path / "padded.pcd"
    with open(p, "wb") as f:
        f.write(b"# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n"
                b"FIELDS x y z _ normal_x normal_y normal_z _ intensity curvature _\n"
                b"SIZE 4 4 4 1 4 4 4 1 4 4 1\nTYPE F F F U F F F U F F U\n"
                b"COUNT 1 1 1 4 1 1 1 4 1 1 8\nWIDTH 4\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\n"
                b"POINTS 4\nDATA binary\n")
        f.write(rec.tobytes())
    got = D.read_pcd(str(p))
    assert set(got) == {"x", "y", "z", "normal_x", "normal_y", "normal_z", "intensity", "curvature"}
    assert np.array_equal(got["x"], np.arange(n)) and np.all(got["normal_z"] == 1) and np.all(got["intensity"] == 7)
    q = tmp_path / "ascii.pcd"
    q.write_text("VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 2\nHEIGHT 1\n"
                 "POINTS 2\nDATA ascii\n1 2 3\n4 5 6\n")
    got = D.read_pcd(str(q))
    assert got["y"].tolist() == [2.0, 5.0]
    r = tmp_path / "bad.pcd"
    r.write_text("VERSION 0.7\nFIELDS x\nSIZE 4\nTYPE F\nCOUNT 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\n"
                 "DATA binary_compressed\n")
    with pytest.raises(ValueError):
        D.read_pcd(str(r))


def test_key_must_equal_line_number(tmp_path):
    d = _small_data()
    D.write_test_data(str(tmp_path), d)
    lines = open(tmp_path / "keyed_poses.csv").read().splitlines()
    open(tmp_path / "keyed_poses.csv", "w").write("\n".join([lines[1], lines[0], lines[2]]) + "\n")
    with pytest.raises(ValueError):
        D.load_test_data(str(tmp_path))


def test_summary_formulas():
    """OutputTestSummary (TestUtils.cc:554-745): errors are the norms of the two halves of
    Logmap(edge.between(gt_from.between(gt_to))), degrees use 180/3.1416."""
    d = _small_data()
    exact = {"key_from": 0, "key_to": 1, "R": np.eye(3), "t": np.array([1.0, 0, 0]), "fitness": 0.01}
    c, s_ = np.cos(0.1), np.sin(0.1)
    off = {"key_from": 2, "key_to": 0, "R": np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]]),
           "t": np.array([-2.0, 0.3, 0]), "fitness": 0.2}
    s = D.summarize(d, [exact, off], [{"key_from": 1, "key_to": 2, "R": np.eye(3),
                                       "t": np.zeros(3), "fitness": 0.5}])
    rows = {r["label"]: r for r in s["rows"]}
    assert rows["a"]["expected"] == 1 and rows["a"]["detected"] == 1 and rows["a"]["recall"] == 1.0
    assert rows["a"]["mean-trans-error(m)"] < 1e-12 and rows["a"]["max-rot-error(deg)"] < 1e-9
    assert rows["a"]["false-positive-rate"] == 1.0
    assert abs(rows["b"]["mean-rot-error(deg)"] - 0.1 * 180 / 3.1416) < 1e-9
    # translation part of the logmap of a pose with rotation 0.1 rad about z and R^T*(-0.3 y)
    assert 0.29 < rows["b"]["mean-trans-error(m)"] < 0.31
    assert s["fp_fitness"][0][0] == 0.5 and abs(s["fp_fitness"][0][1] - 1.0) < 1e-12


@pytest.mark.gpu
def test_evaluate_synthetic_data_set(tmp_path):
    """eval_loop_computation_test.cc end to end on a data set written in the reference's format:
    every true candidate is verified to a few centimetres, no false candidate survives."""
    from lamp_amd import capi
    D.write_test_data(str(tmp_path / "ds"), D.synthetic_test_data())
    data = D.load_test_data(str(tmp_path / "ds"))
    res, fres, info = D.evaluate(data, init_method=capi.INIT_ODOMETRY)
    s = D.summarize(data, res, fres)
    D.write_summary(s, str(tmp_path / "out"), "t")
    assert len(data.fake_candidates) > 0
    assert len(res) == len(data.real_candidates) and len(fres) == 0
    for r in s["rows"]:
        fpr = r["false-positive-rate"]     # 0/0 for labels without false candidates, as in the reference
        assert r["recall"] == 1.0 and (fpr == 0.0 or np.isnan(fpr))
        # neighbour scans are accumulated with the noisy odometry (5 cm, 1 deg): a few centimetres remain
        assert r["max-trans-error(m)"] < 0.15 and r["max-rot-error(deg)"] < 2.0
    assert info["min_ks_size"] == info["max_ks_size"] == 4096
    # the same data set through the evaluation driver's filters (adaptive grid to a target size,
    # random sample): every keyed scan shrinks, loops are still found
    res2, fres2, info2 = D.evaluate(data, init_method=capi.INIT_ODOMETRY,
                                    filters={"adaptive_filter": True, "adaptive_target": 2000,
                                             "adaptive_min_grid": 0.05, "adaptive_max_grid": 0.8,
                                             "random_filter": True, "decimate_percentage": 0.25})
    # (the leaf-size controller is undamped and may stop on a clamp: only "smaller" is certain)
    assert 0 < info2["min_ks_size"] <= info2["max_ks_size"] < 0.75 * 4096 + 1
    assert len(res2) >= 0.75 * len(data.real_candidates) and len(fres2) == 0
    print("filtered keyed-scan sizes", info2)
    head = open(tmp_path / "out" / "t_summary.csv").readline().strip()
    assert head.startswith("label,expected,detected,recall,false-positive-rate,mean-trans-error(m)")
    assert len(open(tmp_path / "out" / "t_fitness.csv").read().splitlines()) == 1 + len(res)
