#!/usr/bin/env python3
"""One C3 list (1,000 pairs x 32,768 points) submitted alone, against the number of parts the
submit is cut into (lgicp_config.reserved[7]; 0 = default: four streams found to overlap)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lamp_amd import capi  # noqa: E402
from lamp_amd.workload import Workload  # noqa: E402

w = Workload("C3", shard=977)
for slots in [int(a) for a in sys.argv[1:]] or [0, 6, 8]:
    cfg = capi.default_config(True, init_method=capi.INIT_IDENTITY)
    cfg.reserved[7] = slots
    ctx = capi.Context(cfg)
    w.register(ctx)
    ctx.wait_registered()
    pk = ctx.pack_candidates(w.candidates())
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        ctx.submit_packed(pk)
        ts.append(1e3 * (time.perf_counter() - t0))
    print(f"parts {slots}: median {np.median(ts[1:]):.2f} ms per list, {1e6 / np.median(ts[1:]):.0f} pairs/s", flush=True)
    ctx.close()
