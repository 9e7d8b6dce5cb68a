#!/usr/bin/env python3
"""One C2 submit at a time on the default context (four parts on four streams), LGICP_TRACE=1: where
the wall time of a lone submit goes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lamp_amd import capi  # noqa: E402
from lamp_amd.workload import Workload  # noqa: E402

work = Workload("C2")
ctx = capi.Context(capi.default_config(True, init_method=capi.INIT_IDENTITY))
work.register(ctx)
pk = ctx.pack_candidates(work.candidates())
for i in range(5):
    t0 = time.perf_counter()
    ctx.submit_packed(pk)
    print(f"submit {i}: wall {1e3 * (time.perf_counter() - t0):.3f} ms", file=sys.stderr, flush=True)
ctx.close()
