#!/usr/bin/env python3
"""Accumulated targets (consecutive keys): wall time and lookup time of a submit against the
first-round radius of the pruned search (lgicp_config.reserved[3], mm; 0 = default 125)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lamp_amd import capi  # noqa: E402
from lamp_amd.workload import Workload  # noqa: E402

consecutive = (sys.argv[1] if len(sys.argv) > 1 else "acc") == "acc"
work = Workload("C2", consecutive_keys=consecutive)
for r0 in (0, 40, 60, 90, 180, 250):
    for timing in (0, 1):
        cfg = capi.default_config(True, init_method=capi.INIT_IDENTITY, stage_timing=timing)
        cfg.reserved[3] = r0
        if timing:
            cfg.reserved[7] = 1
        ctx = capi.Context(cfg)
        work.register(ctx)
        pk = ctx.pack_candidates(work.candidates())
        ts = []
        for i in range(8):
            t0 = time.perf_counter()
            ctx.submit_packed(pk)
            ts.append(1e3 * (time.perf_counter() - t0))
        st = ctx.stats()
        if timing:
            print(f"r0 {r0:3d} mm one stream: nn {st.nn_ms:.2f} reduce {st.reduce_ms:.2f} solve {st.solve_ms:.2f} ms, "
                  f"evaluated distances {st.nn_executed_evals:.3g}", flush=True)
        else:
            print(f"r0 {r0:3d} mm: wall median {np.median(ts[2:]):.3f} ms", flush=True)
        ctx.close()
