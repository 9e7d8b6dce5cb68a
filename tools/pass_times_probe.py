#!/usr/bin/env python3
"""Per-iteration kernel times of one C2 submit on one stream (LGICP_TRACE line), plain keys and
consecutive keys (accumulated targets).  usage: LGICP_TRACE=1 python tools/pass_times_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lamp_amd import capi  # noqa: E402
from lamp_amd.workload import Workload  # noqa: E402

for consecutive in (False, True):
    work = Workload("C2", consecutive_keys=consecutive)
    cfg = capi.default_config(True, init_method=capi.INIT_IDENTITY, stage_timing=1)
    cfg.reserved[7] = 1
    ctx = capi.Context(cfg)
    work.register(ctx)
    pk = ctx.pack_candidates(work.candidates())
    for i in range(3):
        print(f"=== consecutive={consecutive} submit {i}", file=sys.stderr, flush=True)
        ctx.submit_packed(pk)
    ctx.close()
