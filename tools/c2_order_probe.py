"""How much of a C2 step is the scatter of the scan order: the same batch with every keyframe's
points handed over in scan order and in a spatial (Morton) order."""
import os, sys, time
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


for mode in ("scan", "spatial", "scan", "spatial"):
    works = [Workload("C2", shard=s, key_offset=100000 * s) for s in range(4)]
    c = capi.Context(capi.default_config(True, init_method=capi.INIT_IDENTITY))
    pks = []
    for w in works:
        if mode == "spatial":
            for k, (p, n) in w.keyframes.items():
                o = morton(p)
                w.keyframes[k] = (np.ascontiguousarray(p[o]), np.ascontiguousarray(n[o]))
        w.register(c)
        pks.append(c.pack_candidates(list(w.pairs)))
    for i in range(8):
        c.submit_packed(pks[i % 4])
    t0 = time.perf_counter()
    K = 24
    for i in range(K):
        c.submit_packed(pks[i % 4])
    dt = (time.perf_counter() - t0) / K
    s = c.stats()
    print(f"{mode}: {1e3 * dt:.3f} ms per submit", flush=True)
    c.close()
