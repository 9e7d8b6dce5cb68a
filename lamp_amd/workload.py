"""Seeded synthetic loop-closure workloads (VLP-16-like keyframes in procedural rooms).

A keyframe is what a 16-ring lidar (elevations -15..+15 deg in 2 deg steps) sees from a few
sweep positions along a short trajectory inside a box-shaped room with box obstacles: rays are
cast against the axis-aligned geometry, range noise N(0, 0.02 m) is added along the ray, and
the analytic normal of the hit face is stored (flipped toward the sensor).  A candidate pair is
two keyframes of the same place seen from poses that differ by a known SE(3) offset
(rotation <= 5 deg, translation <= 0.5 m).  In configuration C4 (global registration) every
keyframe's own frame is additionally displaced from the pose its odometry reports by a rotation
of up to 45 deg about a random axis and a translation of up to 2.5 m, so the two frames of a pair
differ by up to 90 deg / 5 m while the registered poses know nothing of it: the odometry guess is
wrong by that much and only a feature-based initialisation can recover the pair.  Everything
derives from one integer seed through numpy's PCG64, so the CPU box and the GPU box build
identical clouds.
"""
import numpy as np

SEED_BASE = 0x1A3B0000

# BASELINE.json configs -> (pairs, points per keyframe, distinct keyframes)
CONFIGS = {
    "C1": dict(pairs=1, points=10000, keyframes=2),
    "C2": dict(pairs=256, points=16384, keyframes=64),
    "C3": dict(pairs=1000, points=32768, keyframes=250),
    "C4": dict(pairs=512, points=16384, keyframes=128),
    "C5": dict(pairs=10000, points=65536, keyframes=2000),
}


def _rot(axis, ang):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


def _quat_wxyz(R):
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = [0.0] * 4
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    q = np.array(q)
    return q / np.linalg.norm(q)


def pose_gap(Ta, Tb):
    """(rotation angle in rad, translation norm in m) of Ta^-1 Tb, the angle from sin and cos
    together (arccos of the trace alone cannot resolve small angles on float32 matrices)."""
    dT = np.linalg.inv(np.asarray(Ta, np.float64)) @ np.asarray(Tb, np.float64)
    R = dT[:3, :3]
    s = 0.5 * np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return float(np.arctan2(s, 0.5 * (np.trace(R) - 1.0))), float(np.linalg.norm(dT[:3, 3]))


class Place:
    """A room (20 x 6 x 3 m) with 2-4 box obstacles standing on the floor."""

    def __init__(self, rng):
        self.lo = np.array([-10.0, -3.0, 0.0])
        self.hi = np.array([10.0, 3.0, 3.0])
        self.boxes = []
        for _ in range(int(rng.integers(2, 5))):
            size = np.array([rng.uniform(0.6, 2.5), rng.uniform(0.5, 1.5), rng.uniform(0.5, 2.2)])
            cx = rng.uniform(-8.5, 8.5)
            cy = rng.choice([-1, 1]) * rng.uniform(1.0, 2.2)   # keep the corridor axis free
            lo = np.array([cx - size[0] / 2, cy - size[1] / 2, 0.0])
            self.boxes.append((lo, lo + size))

    def cast(self, origin, dirs):
        """first hit of rays origin + t*dirs: returns t, normal (pointing toward the sensor)"""
        n = len(dirs)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / dirs
        # inside of the room: exit distance
        t1 = (self.lo - origin) * inv
        t2 = (self.hi - origin) * inv
        tfar = np.where(dirs >= 0, t2, t1)
        tfar = np.where(np.isfinite(tfar), tfar, np.inf)
        ax = np.argmin(tfar, axis=1)
        t = tfar[np.arange(n), ax]
        nrm = np.zeros((n, 3))
        nrm[np.arange(n), ax] = -np.sign(dirs[np.arange(n), ax])
        # obstacles: entry distance
        for lo, hi in self.boxes:
            a = (lo - origin) * inv
            b = (hi - origin) * inv
            tn = np.minimum(a, b)
            tx = np.maximum(a, b)
            tn = np.where(np.isnan(tn), -np.inf, tn)
            tx = np.where(np.isnan(tx), np.inf, tx)
            tnear = tn.max(axis=1)
            tfar_b = tx.min(axis=1)
            hit = (tnear < tfar_b) & (tnear > 1e-6) & (tnear < t)
            axb = np.argmax(tn, axis=1)
            t = np.where(hit, tnear, t)
            nb = np.zeros((n, 3))
            nb[np.arange(n), axb] = -np.sign(dirs[np.arange(n), axb])
            nrm = np.where(hit[:, None], nb, nrm)
        return t, nrm


def make_keyframe(place, R_wk, t_wk, n_points, rng, sweeps=4, range_noise=0.02):
    """points/normals in the keyframe frame (pose R_wk, t_wk in the place frame)"""
    elev = np.deg2rad(np.arange(-15, 16, 2, dtype=np.float64))      # 16 rings
    per_sweep = -(-n_points // sweeps)
    n_az = -(-per_sweep // len(elev))
    pts, nrms = [], []
    for s in range(sweeps):
        # sweep origin: along the keyframe's x axis, +-0.3 m
        o_k = np.array([(s - (sweeps - 1) / 2) * 0.2, 0.0, 0.0])
        o_w = R_wk @ o_k + t_wk
        az = (np.arange(n_az) + rng.random()) * (2 * np.pi / n_az)
        A, E = np.meshgrid(az, elev, indexing="ij")
        d_k = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
        d_w = d_k @ R_wk.T
        t, n_w = place.cast(o_w, d_w)
        t = t + range_noise * rng.standard_normal(len(t))
        p_w = o_w + t[:, None] * d_w
        pts.append((p_w - t_wk) @ R_wk)
        nrms.append(n_w @ R_wk)
    pts = np.vstack(pts)
    nrms = np.vstack(nrms)
    keep = rng.permutation(len(pts))[:n_points]
    keep.sort()
    return pts[keep].astype(np.float32), nrms[keep].astype(np.float32)


class Workload:
    """keyframes: dict key -> (xyz, normals); poses: key -> (q_wxyz, t); pairs: list of
    (key_from, key_to); truth: (key_from, key_to) -> 4x4 source->target transform"""

    def __init__(self, config="C2", pairs=None, points=None, keyframes=None, shard=0,
                 pose_noise=(0.0, 0.0), key_offset=0, frame_offset=None, progress=None,
                 consecutive_keys=False):
        """shard: changes the seed (another batch of the same shape); key_offset: added to every
        key, so that several workloads can live in one engine context; frame_offset: (max rotation
        in deg, max translation in m) of every keyframe's frame against its registered pose
        (default: C4 (45, 2.5), other configurations none); progress: called with a line of text
        every 64 keyframes (large configurations take minutes to generate); consecutive_keys: the
        keyframes of one place get consecutive keys, as along a robot's trajectory, so that the
        reference's AccumulateScans (IcpLoopComputation.cc:910-947) finds key +- 1, +- 2 and a
        target grows to up to five scans (default: keys 10 apart, nothing is accumulated)."""
        c = dict(CONFIGS[config])
        if pairs is not None:
            c["pairs"] = pairs
        if points is not None:
            c["points"] = points
        if keyframes is not None:
            c["keyframes"] = keyframes
        self.config, self.n_pairs, self.n_points, self.n_keyframes = config, c["pairs"], c["points"], c["keyframes"]
        cid = {"C1": 1, "C2": 2, "C3": 3, "C4": 4, "C5": 5}[config]
        if frame_offset is None:
            frame_offset = (45.0, 2.5) if config == "C4" else (0.0, 0.0)
        self.frame_offset = frame_offset
        self.seed = SEED_BASE + cid * (1 << 20) + shard * 4099
        rng = np.random.default_rng(self.seed)
        per_place = 8 if self.n_keyframes >= 8 else self.n_keyframes
        n_places = -(-self.n_keyframes // per_place)
        self.keyframes, self.poses, self.true_pose = {}, {}, {}
        place_of = {}
        key = 0
        for pl in range(n_places):
            place = Place(rng)
            base_t = np.array([rng.uniform(-4, 4), rng.uniform(-0.8, 0.8), rng.uniform(0.8, 1.6)])
            base_R = _rot(np.array([0, 0, 1.0]), rng.uniform(-np.pi, np.pi))
            for _ in range(per_place):
                if len(self.keyframes) >= self.n_keyframes:
                    break
                axis = rng.standard_normal(3)
                R = base_R @ _rot(axis, np.deg2rad(rng.uniform(0, 2.5)))
                d = rng.standard_normal(3)
                t = base_t + d / np.linalg.norm(d) * rng.uniform(0, 0.25)
                # keys are spaced by 10 so that no neighbour scan gets accumulated, or consecutive
                # within a place with a gap between places
                k = (1000 + key + 10 * pl if consecutive_keys else 1000 + 10 * key) + key_offset
                key += 1
                pts, nrms = make_keyframe(place, R, t, self.n_points, rng)
                Rs, ts = R, t                      # the sensor's pose = what odometry reports
                if frame_offset[0] > 0 or frame_offset[1] > 0:
                    # the keyframe's own frame = sensor pose * D
                    DR = _rot(rng.standard_normal(3), np.deg2rad(frame_offset[0]) * rng.random())
                    dd = rng.standard_normal(3)
                    Dt = dd / np.linalg.norm(dd) * frame_offset[1] * rng.random()
                    pts = ((pts.astype(np.float64) - Dt) @ DR).astype(np.float32)
                    nrms = (nrms.astype(np.float64) @ DR).astype(np.float32)
                    R, t = Rs @ DR, ts + Rs @ Dt
                self.keyframes[k] = (pts, nrms)
                if progress is not None and len(self.keyframes) % 64 == 0:
                    progress(f"{len(self.keyframes)}/{self.n_keyframes} keyframes generated")
                self.true_pose[k] = (R, t)
                R_reg, t_reg = Rs, ts
                # registered (odometry) pose = truth + optional drift
                Rn = R_reg @ _rot(rng.standard_normal(3), np.deg2rad(pose_noise[0]) * rng.random())
                tn = t_reg + pose_noise[1] * (rng.random(3) * 2 - 1)
                self.poses[k] = (_quat_wxyz(Rn), tn)
                place_of[k] = pl
        keys = sorted(self.keyframes)
        cand = [(a, b) for a in keys for b in keys if a != b and place_of[a] == place_of[b]]
        order = rng.permutation(len(cand))
        reps = -(-self.n_pairs // max(len(cand), 1))
        sel = [cand[i] for _ in range(reps) for i in order][: self.n_pairs]
        self.pairs = sel
        self.truth = {}
        for a, b in set(sel):
            Ra, ta = self.true_pose[a]
            Rb, tb = self.true_pose[b]
            T = np.eye(4)
            T[:3, :3] = Rb.T @ Ra          # source a -> target b
            T[:3, 3] = Rb.T @ (ta - tb)
            self.truth[(a, b)] = T

    def register(self, ctx, batched=True):
        """keyframes and poses into the engine: one lgicp_add_keyframes call (batched), or one
        lgicp_add_keyframe call per scan as the reference's KeyedScanCallback receives them"""
        keys = list(self.keyframes)
        if batched:
            ctx.add_keyframes(keys, [self.keyframes[k][0] for k in keys],
                              [self.keyframes[k][1] for k in keys])
        for k in keys:
            if not batched:
                ctx.add_keyframe(k, *self.keyframes[k])
            ctx.add_pose(k, *self.poses[k])

    def candidates(self):
        return [(a, b) for a, b in self.pairs]
