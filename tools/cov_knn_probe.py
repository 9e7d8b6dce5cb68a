#!/usr/bin/env python3
"""Where the device's k-NN covariances (gicp.hpp:83-155) differ from the oracle's: singular values of
the raw neighbourhood covariance and ties at the k-th neighbour for every differing point."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lamp_amd import capi  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import fixtures as F  # noqa: E402

ctx = capi.Context(init_method=capi.INIT_IDENTITY)
for k in (20, 10):
    xyz, _ = F.random_room_scan(4000, 21)
    got = ctx.covariances_knn(xyz, k, 1e-3)
    ref = O.cov_knn(xyz, k, 1e-3)
    err = np.abs(got - ref).max(axis=(1, 2))
    bad = np.nonzero(err > 1e-9)[0]
    print(f"k={k}: {len(bad)} of {len(xyz)} differ by more than 1e-9; max {err.max():.2e}")
    idx, d2 = O.knn_brute(xyz[bad], xyz, k + 1)
    for r, i in enumerate(bad[:40]):
        pts = xyz[idx[r, :k]].astype(np.float64)
        mean = pts.mean(axis=0)
        c = (pts - mean).T @ (pts - mean) / k
        sv = np.linalg.svd(c, compute_uv=False)
        print(f"  point {i}: err {err[i]:.2e} sv {sv[0]:.3e} {sv[1]:.3e} {sv[2]:.3e} "
              f"gap/s0 {(sv[1] - sv[2]) / sv[0]:.2e} tie_at_k {d2[r, k - 1] == d2[r, k]}")
ctx.close()
