#!/usr/bin/env python3
"""Accumulated targets (consecutive keys, base-station block): wall time of successive submits with
the accumulated-cloud cache on and off.  LGICP_TRACE=1 shows where a submit's time goes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lamp_amd import capi  # noqa: E402
from lamp_amd.workload import Workload  # noqa: E402

work = Workload("C2", consecutive_keys=True)
for cache in (0, 1):
    cfg = capi.default_config(True, init_method=capi.INIT_IDENTITY, accumulated_cache_bytes=cache,
                              stage_timing=1)
    if len(sys.argv) > 1:
        cfg.reserved[7] = int(sys.argv[1])
    ctx = capi.Context(cfg)
    work.register(ctx)
    pk = ctx.pack_candidates(work.candidates())
    ts = []
    for i in range(6):
        t0 = time.perf_counter()
        ctx.submit_packed(pk)
        ts.append(1e3 * (time.perf_counter() - t0))
        st = ctx.stats()
        print(f"cache {'off' if cache else 'on'} submit {i}: wall {ts[-1]:.2f} ms, device span {st.total_ms:.2f}, "
              f"nn {st.nn_ms:.2f} reduce {st.reduce_ms:.2f} solve {st.solve_ms:.2f} prepare {st.cov_ms:.2f}", flush=True)
    ctx.close()
