"""The reference's loop-computation evaluation data set, read, written and evaluated.

Directory layout (loop_closure/src/TestUtils.cc:25-319): `candidates.csv` and
`false_candidates.csv` (key_from,key_to per line), `keyed_poses.csv` and `gt_keyed_poses.csv`
(key,x,y,z,qx,qy,qz,qw,label per line, key == line number), `<key>.pcd` (binary PCD of
pcl::PointXYZINormal).  `evaluate()` follows loop_closure/src/eval_loop_computation_test.cc:
every keyed scan is (optionally filtered,) stripped of its normals and given fresh k-NN normals
(FilterKeyedScans, :210-262), the true candidates and then the false candidates are verified,
and `write_summary()` produces the four CSV files of OutputTestSummary (TestUtils.cc:554-745)
with the same columns and formulas.

The verification itself runs on the GPU engine through lamp_amd.capi; nothing here computes an
alignment.
"""
import os

import numpy as np

_PCD_TYPES = {("F", 4): np.float32, ("F", 8): np.float64, ("U", 1): np.uint8, ("U", 2): np.uint16,
              ("U", 4): np.uint32, ("I", 1): np.int8, ("I", 2): np.int16, ("I", 4): np.int32,
              ("U", 8): np.uint64, ("I", 8): np.int64}


# ------------------------------------------------------------------------------------- PCD
def read_pcd(path):
    """Reads an ascii or binary (uncompressed) PCD file.  Returns a dict of 1-D arrays keyed by
    field name (padding fields `_` are skipped)."""
    with open(path, "rb") as f:
        header = {}
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: no DATA line")
            line = line.decode("ascii", "replace").strip()
            if not line or line.startswith("#"):
                continue
            key, _, rest = line.partition(" ")
            header[key.upper()] = rest.split()
            if key.upper() == "DATA":
                break
        fields = header["FIELDS"]
        sizes = [int(v) for v in header["SIZE"]]
        types = header["TYPE"]
        counts = [int(v) for v in header.get("COUNT", ["1"] * len(fields))]
        n = int(header["POINTS"][0]) if "POINTS" in header else \
            int(header["WIDTH"][0]) * int(header["HEIGHT"][0])
        mode = header["DATA"][0].lower()
        names, formats = [], []
        for i, (name, size, typ, cnt) in enumerate(zip(fields, sizes, types, counts)):
            names.append(name if name != "_" else f"_pad{i}")
            formats.append((_PCD_TYPES[(typ.upper(), size)], (cnt,)) if cnt > 1
                           else _PCD_TYPES[(typ.upper(), size)])
        dtype = np.dtype({"names": names, "formats": formats})
        if mode == "binary":
            rec = np.frombuffer(f.read(n * dtype.itemsize), dtype=dtype, count=n)
        elif mode == "ascii":
            flat = np.loadtxt(f, dtype=np.float64, ndmin=2)
            rec = np.zeros(n, dtype=dtype)
            col = 0
            for name, cnt in zip(names, counts):
                if cnt == 1:
                    rec[name] = flat[:n, col]
                else:
                    rec[name] = flat[:n, col:col + cnt]
                col += cnt
        else:
            raise ValueError(f"{path}: DATA {mode} is not supported (binary or ascii only)")
    return {k: np.array(rec[k]) for k in names if not k.startswith("_pad")}


def write_pcd_binary(path, xyz, normals=None, intensity=None, curvature=None):
    """Writes a binary PCD with the fields of pcl::PointXYZINormal."""
    xyz = np.asarray(xyz, np.float32).reshape(-1, 3)
    n = len(xyz)
    nrm = np.zeros((n, 3), np.float32) if normals is None else np.asarray(normals, np.float32)
    dtype = np.dtype([(k, np.float32) for k in
                      ("x", "y", "z", "intensity", "normal_x", "normal_y", "normal_z", "curvature")])
    rec = np.zeros(n, dtype)
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    rec["normal_x"], rec["normal_y"], rec["normal_z"] = nrm[:, 0], nrm[:, 1], nrm[:, 2]
    if intensity is not None:
        rec["intensity"] = intensity
    if curvature is not None:
        rec["curvature"] = curvature
    with open(path, "wb") as f:
        f.write(("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n"
                 "FIELDS x y z intensity normal_x normal_y normal_z curvature\n"
                 "SIZE 4 4 4 4 4 4 4 4\nTYPE F F F F F F F F\nCOUNT 1 1 1 1 1 1 1 1\n"
                 f"WIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n").encode())
        f.write(rec.tobytes())


# ------------------------------------------------------------------------------------- CSV
def _read_candidates(path):
    out = []
    with open(path) as f:
        for line in f:
            parts = line.strip().split(",")
            if len(parts) >= 2 and parts[0] != "":
                out.append((int(parts[0]), int(parts[1])))
    return out


def _read_poses(path):
    """-> list of (q_wxyz, t), list of labels; the key must equal the line number
    (TestUtils.cc:128-133)."""
    poses, labels = [], []
    with open(path) as f:
        for line_num, line in enumerate(f):
            parts = line.rstrip("\n").split(",")
            if len(parts) < 8:
                continue
            if int(float(parts[0])) != line_num:
                raise ValueError(f"{path}: key {parts[0]} on line {line_num}")
            x, y, z, qx, qy, qz, qw = (float(v) for v in parts[1:8])
            poses.append(((qw, qx, qy, qz), (x, y, z)))
            labels.append(parts[8] if len(parts) > 8 else "")
    return poses, labels


def _write_poses(path, poses, labels):
    with open(path, "w") as f:
        for i, ((qw, qx, qy, qz), (x, y, z)) in enumerate(poses):
            vals = ",".join(repr(float(v)) for v in (x, y, z, qx, qy, qz, qw))
            f.write(f"{i},{vals},{labels[i]}\n")


class TestData:
    """test_utils::TestData (loop_closure/include/loop_closure/TestUtils.h)."""
    __test__ = False

    def __init__(self):
        self.real_candidates, self.fake_candidates = [], []
        self.odom_poses, self.gt_poses, self.labels = [], [], []
        self.scans = []          # per key: dict with "xyz" [n,3] and "normals" [n,3]


def load_test_data(path):
    """LoadExistingTestData (TestUtils.cc:25-60)."""
    d = TestData()
    d.real_candidates = _read_candidates(os.path.join(path, "candidates.csv"))
    d.fake_candidates = _read_candidates(os.path.join(path, "false_candidates.csv"))
    d.odom_poses, d.labels = _read_poses(os.path.join(path, "keyed_poses.csv"))
    d.gt_poses, _ = _read_poses(os.path.join(path, "gt_keyed_poses.csv"))
    for key in range(len(d.odom_poses)):
        rec = read_pcd(os.path.join(path, f"{key}.pcd"))
        xyz = np.stack([rec["x"], rec["y"], rec["z"]], 1).astype(np.float32)
        if "normal_x" in rec:
            nrm = np.stack([rec["normal_x"], rec["normal_y"], rec["normal_z"]], 1).astype(np.float32)
        else:
            nrm = np.zeros_like(xyz)
        d.scans.append({"xyz": xyz, "normals": nrm,
                        "intensity": rec.get("intensity", np.zeros(len(xyz), np.float32))})
    return d


def write_test_data(path, d):
    """WriteTestDataToFile (TestUtils.cc:271-319)."""
    os.makedirs(path, exist_ok=True)
    for name, cands in (("candidates.csv", d.real_candidates),
                        ("false_candidates.csv", d.fake_candidates)):
        with open(os.path.join(path, name), "w") as f:
            for a, b in cands:
                f.write(f"{a},{b}\n")
    _write_poses(os.path.join(path, "gt_keyed_poses.csv"), d.gt_poses, d.labels)
    _write_poses(os.path.join(path, "keyed_poses.csv"), d.odom_poses, d.labels)
    for key, s in enumerate(d.scans):
        write_pcd_binary(os.path.join(path, f"{key}.pcd"), s["xyz"], s.get("normals"),
                         s.get("intensity"))


# ------------------------------------------------------------------------------------- filters
def adaptive_filter(ctx, xyz, normals, target, min_leaf, max_leaf, observability_check=False,
                    init_leaf=0.25):
    """EvalIcpLoopCompute::AdaptiveFilter (eval_loop_computation_test.cc:127-174): voxel-grid the
    original cloud with a leaf size that is steered until the output is within 10 points of the
    target (at most 100 rounds).  The grid and the observability run on the device."""
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    normals = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    if len(xyz) < target:
        return xyz, normals
    leaf, out, out_n = init_leaf, xyz, normals
    prev_obs = 0.0
    if observability_check:
        prev_obs = float(np.min(ctx.icp_observability(out, out_n))) / len(out)
    for _ in range(100):
        if abs(len(out) - target) <= 10:
            break
        out, out_n = ctx.voxel_grid(xyz, leaf, normals)
        obs_factor = 0.0
        if observability_check:
            obs = float(np.min(ctx.icp_observability(out, out_n))) / len(out)
            with np.errstate(divide="ignore", invalid="ignore"):
                obs_factor = float((np.float64(prev_obs) - obs) / np.float64(prev_obs))
            prev_obs = obs
        size_factor = len(out) / float(target)
        v = leaf * (size_factor - abs(size_factor - 1) * obs_factor)
        m = v if min_leaf < v else min_leaf      # std::max(min_leaf, v), NaN included
        leaf = m if m < max_leaf else max_leaf   # std::min(max_leaf, m)
    return out, out_n


def filter_keyed_scan(ctx, xyz, normals, filters, seed=0):
    """EvalIcpLoopCompute::FilterKeyedScans up to the point where the normals are dropped
    (eval_loop_computation_test.cc:211-250): adaptive grid, random sample, fixed grid."""
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    if normals is None:
        normals = np.zeros_like(xyz)
    if filters.get("adaptive_filter"):
        xyz, normals = adaptive_filter(ctx, xyz, normals, filters.get("adaptive_target", 3000),
                                       filters.get("adaptive_min_grid", 0.1),
                                       filters.get("adaptive_max_grid", 0.8),
                                       filters.get("observability_check", False))
    if filters.get("random_filter"):
        keep = int((1.0 - filters.get("decimate_percentage", 0.8)) * len(xyz))
        xyz, normals = ctx.random_sample(xyz, keep, seed, normals)
    if filters.get("grid_filter"):
        xyz, normals = ctx.voxel_grid(xyz, filters.get("grid_res", 0.8), normals)
    return xyz, normals


# ------------------------------------------------------------------------------------- poses
def _R_from_q(q):
    w, x, y, z = q
    n = np.sqrt(w * w + x * x + y * y + z * z)
    w, x, y, z = w / n, x / n, y / n, z / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _between(pa, pb):
    """gtsam Pose3::between: a^-1 * b, poses as (R, t)."""
    Ra, ta = pa
    Rb, tb = pb
    return Ra.T @ Rb, Ra.T @ (tb - ta)


def _logmap(R, t):
    """gtsam Pose3::Logmap -> (omega, u)."""
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    W = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-10:
        w = 0.5 * W
    else:
        w = th / (2 * np.sin(th)) * W
    th = np.linalg.norm(w)
    if th < 1e-10:
        return w, t.copy()
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    Vinv = np.eye(3) - 0.5 * K + (1 / th ** 2) * (1 - th * np.sin(th) / (2 * (1 - np.cos(th)))) * K @ K
    return w, Vinv @ t


# ------------------------------------------------------------------------------------- evaluation
def evaluate(data, use_gt_odom=False, filters=None, normals_k=10, **config):
    """Runs both candidate lists through the engine.  Returns (results, false_results): lists of
    dicts with key_from, key_to, R, t (the loop-closure edge) and fitness, one per accepted
    candidate — what IcpLoopComputation leaves in its output queue."""
    from . import capi
    filters = filters or {}
    ctx = capi.Context(**config)
    try:
        sizes, clouds, normals = [], [], []
        for key, s in enumerate(data.scans):
            xyz, _ = filter_keyed_scan(ctx, s["xyz"], s.get("normals"), filters, seed=key)
            # ConvertPointCloud + AddNormals: the stored normals are dropped and recomputed
            nrm, _ = ctx.compute_normals(xyz, normals_k) if len(xyz) else (np.zeros((0, 3)), None)
            clouds.append(np.ascontiguousarray(xyz, np.float32))
            normals.append(np.ascontiguousarray(nrm, np.float32))
            sizes.append(len(xyz))
        # one registration call for the whole data set: index, covariances and constants of all
        # scans are built together on the device
        ctx.add_keyframes(list(range(len(clouds))), clouds, normals)
        for key, (q, t) in enumerate(data.gt_poses if use_gt_odom else data.odom_poses):
            ctx.add_pose(key, q, t)

        def run(cands):
            out = []
            for r in ctx.submit(list(cands)):
                if r.status == capi.PAIR_OK:
                    out.append({"key_from": r.key_from, "key_to": r.key_to,
                                "R": np.array(r.delta_R).reshape(3, 3), "t": np.array(r.delta_t),
                                "fitness": r.fitness})
            return out

        results = run(data.real_candidates)
        false_results = run(data.fake_candidates)
    finally:
        ctx.close()
    return results, false_results, {"max_ks_size": max(sizes, default=0),
                                    "min_ks_size": min(sizes, default=0)}


def summarize(data, results, false_results):
    """The per-label table and the raw error lists of OutputTestSummary."""
    gt = [(_R_from_q(q), np.array(t, np.float64)) for q, t in data.gt_poses]

    def errors(edge):
        Rg, tg = _between(gt[edge["key_from"]], gt[edge["key_to"]])
        Re, te = _between((edge["R"], edge["t"]), (Rg, tg))
        w, u = _logmap(Re, te)
        return float(np.linalg.norm(u)), float(np.linalg.norm(w))

    expected, false_expected = {}, {}
    for a, _ in data.real_candidates:
        expected[data.labels[a]] = expected.get(data.labels[a], 0) + 1
    for a, _ in data.fake_candidates:
        false_expected[data.labels[a]] = false_expected.get(data.labels[a], 0) + 1
    per = {l: {"detected": 0, "false": 0, "trans": [], "rot": []} for l in expected}
    fit, fit_fp = [], []
    for e in results:
        l = data.labels[e["key_from"]]
        te, re = errors(e)
        per.setdefault(l, {"detected": 0, "false": 0, "trans": [], "rot": []})
        per[l]["detected"] += 1
        per[l]["trans"].append(te)
        per[l]["rot"].append(re)
        fit.append((e["fitness"], te, re))
    for e in false_results:
        l = data.labels[e["key_from"]]
        te, re = errors(e)
        per.setdefault(l, {"detected": 0, "false": 0, "trans": [], "rot": []})
        per[l]["false"] += 1
        fit_fp.append((e["fitness"], te, re))
    deg = 180 / 3.1416      # the reference's constant
    rows = []
    for l, n_exp in expected.items():
        p = per[l]
        n = p["detected"]
        rows.append({
            "label": l, "expected": n_exp, "detected": n, "recall": n / n_exp,
            "false-positive-rate": p["false"] / false_expected[l] if false_expected.get(l) else float("nan"),
            "mean-trans-error(m)": sum(p["trans"]) / n if n else float("nan"),
            "min-trans-error(m)": min(p["trans"]) if n else float("inf"),
            "max-trans-error(m)": max(p["trans"]) if n else 0.0,
            "mean-rot-error(deg)": sum(p["rot"]) / n * deg if n else float("nan"),
            "min-rot-error(deg)": min(p["rot"]) * deg if n else float("inf"),
            "max-rot-error(deg)": max(p["rot"]) * deg if n else 0.0,
        })
    return {"rows": rows, "fitness": fit, "fp_fitness": fit_fp, "per_label": per, "deg": deg}


def write_summary(summary, output_dir, test_name):
    """<test_name>_summary.csv, _fitness.csv, _fp_fitness.csv, _error.csv."""
    os.makedirs(output_dir, exist_ok=True)
    cols = ["label", "expected", "detected", "recall", "false-positive-rate", "mean-trans-error(m)",
            "min-trans-error(m)", "max-trans-error(m)", "mean-rot-error(deg)", "min-rot-error(deg)",
            "max-rot-error(deg)"]
    with open(os.path.join(output_dir, f"{test_name}_summary.csv"), "w") as f:
        f.write(",".join(cols) + "\n")
        for r in summary["rows"]:
            f.write(",".join(str(r[c]) for c in cols) + "\n")
    deg = summary["deg"]
    for name, rows in (("fitness", summary["fitness"]), ("fp_fitness", summary["fp_fitness"])):
        with open(os.path.join(output_dir, f"{test_name}_{name}.csv"), "w") as f:
            f.write("fitness,trans-error(m),rot-error(deg)\n")
            for fitv, te, re in rows:
                f.write(f"{fitv},{te},{re * deg}\n")
    with open(os.path.join(output_dir, f"{test_name}_error.csv"), "w") as f:
        f.write("label,trans-error(m),rot-error(deg)\n")
        for l, p in summary["per_label"].items():
            for te, re in zip(p["trans"], p["rot"]):
                f.write(f"{l},{te},{re * deg}\n")


def synthetic_test_data(n_places=4, per_place=4, points=4096, seed=5, odom_noise=(1.0, 0.05)):
    """A small data set in the reference's format from the bench's scene generator: true
    candidates pair keyframes of one place, false candidates pair keyframes of different places
    (their scans do not overlap)."""
    from .workload import Place, make_keyframe, _rot, _quat_wxyz
    rng = np.random.default_rng(seed)
    d = TestData()
    place_of = []
    for pl in range(n_places):
        place = Place(rng)
        base_t = np.array([rng.uniform(-4, 4), rng.uniform(-0.8, 0.8), rng.uniform(0.8, 1.6)])
        base_R = _rot(np.array([0, 0, 1.0]), rng.uniform(-np.pi, np.pi))
        for _ in range(per_place):
            R = base_R @ _rot(rng.standard_normal(3), np.deg2rad(rng.uniform(0, 2.5)))
            dv = rng.standard_normal(3)
            t = base_t + dv / np.linalg.norm(dv) * rng.uniform(0, 0.25) + np.array([100.0 * pl, 0, 0])
            xyz, nrm = make_keyframe(place, R, t - np.array([100.0 * pl, 0, 0]), points, rng)
            d.scans.append({"xyz": xyz, "normals": nrm})
            d.gt_poses.append((tuple(_quat_wxyz(R)), tuple(t)))
            Rn = R @ _rot(rng.standard_normal(3), np.deg2rad(odom_noise[0]) * rng.random())
            tn = t + odom_noise[1] * (rng.random(3) * 2 - 1)
            d.odom_poses.append((tuple(_quat_wxyz(Rn)), tuple(tn)))
            d.labels.append(f"place{pl}")
            place_of.append(pl)
    keys = range(len(d.scans))
    d.real_candidates = [(a, b) for a in keys for b in keys if a < b and place_of[a] == place_of[b]]
    # places at least two apart: the scans accumulated around either key (its +-2 neighbours,
    # IcpLoopComputation.cc:910-947) then never reach into the other key's place
    d.fake_candidates = [(a, b) for a in keys for b in keys
                         if a < b and abs(place_of[a] - place_of[b]) >= 2][: len(d.real_candidates)]
    return d


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    ev = sub.add_parser("eval", help="evaluate a data set directory")
    ev.add_argument("dataset_path")
    ev.add_argument("output_dir")
    ev.add_argument("--test-name", default="lamp_amd")
    ev.add_argument("--use-gt-odom", action="store_true")
    ev.add_argument("--solver", default="gn", choices=["gn", "bfgs"],
                    help="inner solver: Gauss-Newton (default) or the reference's BFGS (lgicp_config.solver)")
    ev.add_argument("--robot", action="store_true",
                    help="the reference's robot parameter block instead of the base station's")
    sy = sub.add_parser("synth", help="write a synthetic data set in the reference's format")
    sy.add_argument("output_dir")
    args = ap.parse_args(argv)
    if args.cmd == "synth":
        write_test_data(args.output_dir, synthetic_test_data())
        return 0
    from . import capi
    data = load_test_data(args.dataset_path)
    res, fres, info = evaluate(data, use_gt_odom=args.use_gt_odom, init_method=capi.INIT_ODOMETRY,
                               base_station=not args.robot,
                               solver=capi.SOLVER_BFGS if args.solver == "bfgs" else capi.SOLVER_GAUSS_NEWTON)
    s = summarize(data, res, fres)
    write_summary(s, args.output_dir, args.test_name)
    print(f"{len(res)} loop closures, {len(fres)} incorrect loop closures, "
          f"keyed scan sizes {info['min_ks_size']}..{info['max_ks_size']}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
