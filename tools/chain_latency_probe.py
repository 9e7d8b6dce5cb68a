#!/usr/bin/env python3
"""Latency of one outer iteration when nothing else runs: the straggler of a C2 batch (the pair
with the most iterations) submitted alone, and the 1, 2, 4, 8, 16 pairs with the most iterations
together.  wall / iterations of the longest pair = what one iteration of a part's tail costs."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lamp_amd import capi  # noqa: E402
from lamp_amd.workload import Workload  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctx = capi.Context(capi.default_config(True, init_method=capi.INIT_IDENTITY))
w = Workload("C2", shard=b, key_offset=b * 100000)
w.register(ctx)
ctx.wait_registered()
cands = w.candidates()
out = ctx.submit_packed(ctx.pack_candidates(cands))
its = np.array([out[k].iterations for k in range(len(cands))])
order = np.argsort(-its)
for n in (1, 2, 4, 8, 16, 64):
    sel = [cands[i] for i in order[:n]]
    pk = ctx.pack_candidates(sel)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        ctx.submit_packed(pk)
        ts.append(1e3 * (time.perf_counter() - t0))
    print(f"{n:3d} pairs, iterations {its[order[0]]}..{its[order[n - 1]]}: wall {np.median(ts):.3f} ms = "
          f"{1e3 * np.median(ts) / its[order[0]]:.1f} us per iteration of the longest pair", flush=True)
ctx.close()
