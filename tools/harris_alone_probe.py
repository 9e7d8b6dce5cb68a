#!/usr/bin/env python3
"""Harris keypoints of one 16k-point C4 keyframe, ten times in a row, nothing else on the GPU: run
under rocprofv3 --kernel-trace --stats for the kernels' durations when they are alone."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lamp_amd import capi  # noqa: E402
if os.environ.get("LGICP_LIB"):   # another build of the library (A/B comparisons)
    capi.LIB_PATH = os.path.abspath(os.environ["LGICP_LIB"])
from lamp_amd.workload import Workload  # noqa: E402

w = Workload("C4", pairs=4, keyframes=4)
xyz, nrm = w.keyframes[sorted(w.keyframes)[0]]
res = [0] * 8
res[5] = int(os.environ.get("PROBE_R5", "0"))
c = capi.Context(init_method=capi.INIT_TEASERPP, harris_radius=0.5, reserved=(capi.C.c_int * 8)(*res))
for _ in range(10):
    idx = c.harris_keypoints(xyz, nrm)
print(len(idx), "keypoints")
c.close()
