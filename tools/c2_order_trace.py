"""Per-iteration kernel times (LGICP_TRACE=1) of a C2 batch with every keyframe's points handed
over in scan order or in a spatial (Morton) order: argument scan|spatial"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from lamp_amd import capi
from lamp_amd.workload import Workload


def morton(p):
    q = ((p - p.min(0)) / (np.ptp(p, 0).max() + 1e-9) * 1023).astype(np.uint64)
    code = np.zeros(len(p), np.uint64)
    for b in range(10):
        for a in range(3):
            code |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return np.argsort(code, kind="stable")


mode = sys.argv[1] if len(sys.argv) > 1 else "scan"
w = Workload("C2")
if mode == "spatial":
    for k, (p, n) in w.keyframes.items():
        o = morton(p)
        w.keyframes[k] = (np.ascontiguousarray(p[o]), np.ascontiguousarray(n[o]))
cfg = capi.default_config(True, init_method=capi.INIT_IDENTITY)
cfg.reserved[7] = 1
c = capi.Context(cfg)
w.register(c)
for _ in range(3):
    c.submit(w.candidates())
c.close()
