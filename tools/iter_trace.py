#!/usr/bin/env python3
"""Per-iteration kernel times of one C2 submit on one stream (run with LGICP_TRACE=1: the library
prints nn/reduce/solve microseconds per outer iteration on stderr)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lamp_amd import capi  # noqa: E402
if os.environ.get("LGICP_LIB"):   # another build of the library (A/B comparisons)
    capi.LIB_PATH = os.path.abspath(os.environ["LGICP_LIB"])
from lamp_amd.workload import Workload  # noqa: E402

cfg_name = sys.argv[1] if len(sys.argv) > 1 else "C2"
work = Workload(cfg_name)
cfg = capi.default_config(True, init_method=capi.INIT_IDENTITY)
cfg.reserved[7] = 1
ctx = capi.Context(cfg)
work.register(ctx)
for _ in range(3):
    ctx.submit(work.candidates())
ctx.close()
