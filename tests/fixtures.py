"""Point-cloud fixtures with the shapes the reference's tests use
(loop_closure/test/test_artifacts.h:14-87): a 10x10 plane, a 3-plane corner and a box,
plus seeded synthetic scans for the larger parity cases."""
import numpy as np


def generate_plane(x_points=10, y_points=10, step_x=0.1, step_y=0.1):
    """test_artifacts.h:14-31 — points (ix*step, iy*step, 0), normal (0,0,1)."""
    sx, sy = np.float32(step_x), np.float32(step_y)
    pts = [(np.float32(ix) * sx, np.float32(iy) * sy, np.float32(0))
           for ix in range(x_points) for iy in range(y_points)]
    xyz = np.array(pts, np.float32)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (len(xyz), 1))
    return xyz, nrm


def _transform_with_normals(xyz, nrm, T):
    """pcl::transformPointCloudWithNormals on exact 0/+-1 matrices (no rounding involved)."""
    T = np.asarray(T, np.float32)
    return (xyz @ T[:3, :3].T + T[:3, 3]).astype(np.float32), (nrm @ T[:3, :3].T).astype(np.float32)


def generate_corner():
    """test_artifacts.h:33-59 — planes z=0 (n=+z), y=0 (n=-y), x=0 (n=-x); 300 points."""
    xyz, nrm = generate_plane()
    T1 = np.zeros((4, 4), np.float32)
    T1[0, 0], T1[1, 2], T1[2, 1], T1[3, 3] = 1, -1, 1, 1
    T2 = np.zeros((4, 4), np.float32)
    T2[0, 2], T2[1, 1], T2[2, 0], T2[3, 3] = -1, 1, 1, 1
    x1, n1 = _transform_with_normals(xyz, nrm, T1)
    x2, n2 = _transform_with_normals(xyz, nrm, T2)
    return np.vstack([xyz, x1, x2]), np.vstack([nrm, n1, n2])


def generate_box():
    """test_artifacts.h:61-87 — corner + its point mirror moved to (1,1,1); 600 points."""
    xyz, nrm = generate_corner()
    T = np.zeros((4, 4), np.float32)
    T[0, 0], T[1, 1], T[2, 2], T[3, 3] = -1, 1, -1, 1
    x1, n1 = _transform_with_normals(xyz, nrm, T)
    T = np.zeros((4, 4), np.float32)
    T[0, 1], T[1, 0], T[2, 2], T[3, 3] = -1, 1, 1, 1
    T[0, 3] = T[1, 3] = T[2, 3] = 1
    x1, n1 = _transform_with_normals(x1, n1, T)
    return np.vstack([xyz, x1]), np.vstack([nrm, n1])


def shifted_corner():
    """corner moved by T = (1, -0.001, 0) as in test_loop_computation.cc:133-137."""
    xyz, nrm = generate_corner()
    T = np.eye(4, dtype=np.float32)
    T[0, 3] = 1
    T[1, 3] = -0.001
    moved = np.empty_like(xyz)
    # pcl::transformPointCloudWithNormals arithmetic: ((m00*x + m01*y) + m02*z) + m03 in float
    for r in range(3):
        moved[:, r] = ((T[r, 0] * xyz[:, 0] + T[r, 1] * xyz[:, 1]) + T[r, 2] * xyz[:, 2]) + T[r, 3]
    return moved, nrm.copy(), T


def rot_zyx(roll, pitch, yaw):
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def random_room_scan(n, seed, noise=0.01):
    """Seeded synthetic scan: points on the inside faces of a 12 x 8 x 3 m room with two
    box obstacles, analytic normals pointing into free space."""
    rng = np.random.default_rng(seed)
    faces = []  # (origin, u, v, normal)
    L, W, H = 12.0, 8.0, 3.0
    faces += [((-L / 2, -W / 2, 0), (L, 0, 0), (0, W, 0), (0, 0, 1)),
              ((-L / 2, -W / 2, H), (L, 0, 0), (0, W, 0), (0, 0, -1)),
              ((-L / 2, -W / 2, 0), (L, 0, 0), (0, 0, H), (0, 1, 0)),
              ((-L / 2, W / 2, 0), (L, 0, 0), (0, 0, H), (0, -1, 0)),
              ((-L / 2, -W / 2, 0), (0, W, 0), (0, 0, H), (1, 0, 0)),
              ((L / 2, -W / 2, 0), (0, W, 0), (0, 0, H), (-1, 0, 0))]
    for (cx, cy, sx, sy, sz) in [(2.0, 1.0, 1.0, 1.5, 1.2), (-3.0, -2.0, 2.0, 1.0, 2.0)]:
        faces += [((cx, cy, sz), (sx, 0, 0), (0, sy, 0), (0, 0, 1)),
                  ((cx, cy, 0), (sx, 0, 0), (0, 0, sz), (0, -1, 0)),
                  ((cx, cy + sy, 0), (sx, 0, 0), (0, 0, sz), (0, 1, 0)),
                  ((cx, cy, 0), (0, sy, 0), (0, 0, sz), (-1, 0, 0)),
                  ((cx + sx, cy, 0), (0, sy, 0), (0, 0, sz), (1, 0, 0))]
    areas = np.array([np.linalg.norm(np.cross(f[1], f[2])) for f in faces])
    which = rng.choice(len(faces), size=n, p=areas / areas.sum())
    a, b = rng.random(n), rng.random(n)
    o = np.array([faces[w][0] for w in which])
    u = np.array([faces[w][1] for w in which])
    v = np.array([faces[w][2] for w in which])
    nr = np.array([faces[w][3] for w in which], np.float64)
    xyz = o + a[:, None] * u + b[:, None] * v + noise * rng.standard_normal(n)[:, None] * nr
    return xyz.astype(np.float32), nr.astype(np.float32)


def make_pair(n, seed, rot_deg=3.0, trans=0.3, noise=0.01):
    """target = scan A; source = an independent sampling of the same room expressed in a
    frame offset by a known SE(3) (so T_true maps source -> target)."""
    rng = np.random.default_rng(seed + 7919)
    tgt, tgt_n = random_room_scan(n, seed, noise)
    src_w, src_nw = random_room_scan(n, seed + 104729, noise)
    ang = np.deg2rad(rot_deg) * (rng.random(3) * 2 - 1)
    R = rot_zyx(*ang)
    t = trans * (rng.random(3) * 2 - 1)
    # p_tgt = R p_src + t  ->  p_src = R^T (p_w - t)
    src = ((src_w.astype(np.float64) - t) @ R).astype(np.float32)
    src_n = (src_nw.astype(np.float64) @ R).astype(np.float32)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    return src, src_n, tgt, tgt_n, T
