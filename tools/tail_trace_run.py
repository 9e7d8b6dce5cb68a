#!/usr/bin/env python3
"""Submits one C2 batch (bench batch index argv[1], default 2: the one with the longest straggler
tail) a few times, 30 ms apart, for a kernel trace (tools/tail_trace.py reads it)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lamp_amd import capi  # noqa: E402
from lamp_amd.workload import Workload  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctx = capi.Context(capi.default_config(True, init_method=capi.INIT_IDENTITY))
w = Workload("C2", shard=b, key_offset=b * 100000)
w.register(ctx)
ctx.wait_registered()
pk = ctx.pack_candidates(w.candidates())
for i in range(4):
    time.sleep(0.03)
    t0 = time.perf_counter()
    ctx.submit_packed(pk)
    print(f"submit {i}: {1e3 * (time.perf_counter() - t0):.3f} ms", flush=True)
ctx.close()
