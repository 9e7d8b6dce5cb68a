#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace (csv) of tools/tail_trace_run.py: takes the last burst of kernels
(bursts are separated by > 10 ms), and prints per stream the kernels of its last `n` launches with
start offset, duration and the gap to the previous kernel's end on the same stream.
Usage: python tools/tail_trace.py trace.csv [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 45
skey = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"


def short(n):
    for k in ("nn_pruned", "gicp_reduce", "gicp_solve", "post_reduce", "copy16", "guess_transform"):
        if k in n:
            return k
    return n.split("(")[0][-28:]


ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[skey], short(r["Kernel_Name"]),
             int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)) for r in rows)
bursts, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in cur) > 10_000_000:
        bursts.append(cur)
        cur = []
    cur.append(e)
bursts.append(cur)
last = bursts[-1]
t0 = min(e[0] for e in last)
print(f"{len(bursts)} bursts; last: {len(last)} kernels, {(max(e[1] for e in last) - t0) / 1e3:.1f} us")
streams = {}
for e in last:
    streams.setdefault(e[2], []).append(e)
for s, lst in sorted(streams.items()):
    print(f"--- stream {s}: {len(lst)} kernels, ends at {(lst[-1][1] - t0) / 1e3:.1f} us")
    prev_end = None
    for i, (a, b, _, name, grid) in enumerate(lst):
        if i >= len(lst) - n_last:
            gap = (a - prev_end) / 1e3 if prev_end is not None else 0.0
            print(f"  {(a - t0) / 1e3:9.1f} us  {name:16s} dur {(b - a) / 1e3:7.1f}  gap {gap:6.1f}  grid {grid}")
        prev_end = b
