#!/usr/bin/env python3
"""Lone C2 submits: each of the bench's four batches submitted over and over, then the four in
rotation.  Tells apart what a batch's own iteration counts cost from what meeting cold caches costs.
Usage: python tools/batch_variance_probe.py [repeats]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lamp_amd import capi  # noqa: E402
from lamp_amd.workload import Workload  # noqa: E402

KEY_STRIDE = 100000
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cfg = capi.default_config(True, init_method=capi.INIT_IDENTITY)
cfg.reserved[5] = int(os.environ.get("PROBE_R5", "0"))      # loop variants (lamp_gicp.h)
cfg.reserved[7] = int(os.environ.get("PROBE_SLOTS", "0"))   # parts of a submit side by side
ctx = capi.Context(cfg)
packed, its = [], []
for b in range(4):
    w = Workload("C2", shard=b, key_offset=b * KEY_STRIDE)
    w.register(ctx)
    packed.append(ctx.pack_candidates(w.candidates()))
ctx.wait_registered()
for b in range(4):
    out = ctx.submit_packed(packed[b])
    it = np.array([r.iterations for r in out])
    its.append(it)
for b in range(4):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx.submit_packed(packed[b])
        ts.append(1e3 * (time.perf_counter() - t0))
    it = its[b]
    parts = [it[i * 64:(i + 1) * 64] for i in range(4)]
    print(f"batch {b} alone: median {np.median(ts):.3f} ms min {min(ts):.3f} | iterations mean "
          f"{it.mean():.2f} max {it.max()} sum {it.sum()} | per part max {[int(p.max()) for p in parts]} "
          f"sum {[int(p.sum()) for p in parts]}", flush=True)
ts = []
for i in range(4 * reps):
    t0 = time.perf_counter()
    ctx.submit_packed(packed[i % 4])
    ts.append(1e3 * (time.perf_counter() - t0))
ts = np.array(ts)
print("rotating: median %.3f ms | per batch %s" % (np.median(ts), [round(float(np.median(ts[b::4])), 3) for b in range(4)]))
ctx.close()
