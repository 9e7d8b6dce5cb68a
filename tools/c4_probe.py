#!/usr/bin/env python3
"""Probe of configuration C4 (large initial misalignment): which Harris radius lets the feature
front end recover the pairs, with TEASER++ and with SAC-IA.  Prints one line per setting."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lamp_amd import capi  # noqa: E402
from lamp_amd.workload import Workload  # noqa: E402
from oracle import oracle as O  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
kfs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
work = Workload("C4", pairs=pairs, keyframes=kfs)
for method, name in ((capi.INIT_TEASERPP, "teaser"), (capi.INIT_FEATURES, "sac")):
    for radius in (0.1, 0.3, 0.5, 1.0):
        cfg = capi.default_config(True, init_method=method, harris_radius=radius,
                                  transform_thresholding=0)
        ctx = capi.Context(cfg)
        try:
            work.register(ctx)
            t0 = time.perf_counter()
            res = ctx.submit(work.candidates())
            t1 = time.perf_counter()
            res2 = ctx.submit(work.candidates())
            t2 = time.perf_counter()
        finally:
            ctx.close()
        ok = good = 0
        errs = []
        for r, (a, b) in zip(res, work.pairs):
            ang, te = O.pose_gap(work.truth[(a, b)], np.array(r.T_icp, np.float64).reshape(4, 4))
            errs.append((ang, te))
            ok += r.status == 0
            good += ang < 0.01 and te < 0.05
        st = [r.status for r in res]
        print(f"{name} radius {radius}: status0 {ok}/{len(res)}, within 0.01 rad / 5 cm of truth {good}, "
              f"init_fitness median {np.median([r.init_fitness for r in res]):.1f}, "
              f"first submit {1e3 * (t1 - t0):.0f} ms, second {1e3 * (t2 - t1):.0f} ms, "
              f"statuses {sorted(set(st))}", flush=True)
