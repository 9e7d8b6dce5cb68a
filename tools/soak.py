"""Submits the same C2 batch many times: results must not drift, host and device memory must not grow."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import psutil
from lamp_amd import capi, workload


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    wl = workload.Workload("C2")
    c = capi.Context(init_method=capi.INIT_IDENTITY)
    wl.register(c)
    packed = c.pack_candidates(wl.candidates())
    proc = psutil.Process()
    first = None
    t0 = time.time()
    for k in range(n):
        out = c.submit_packed(packed)
        T = np.array([list(out[i].T_icp) for i in range(0, packed[1], 17)])
        if first is None:
            first, rss0 = T, proc.memory_info().rss
        elif not np.array_equal(T, first):
            print(json.dumps({"submit": k, "error": "results changed"}), flush=True)
            sys.exit(1)
        if k % 250 == 0:
            print(json.dumps({"submit": k, "rss_mb": proc.memory_info().rss / 2**20,
                              "elapsed_s": time.time() - t0}), flush=True)
    print(json.dumps({"submits": n, "rss_growth_mb": (proc.memory_info().rss - rss0) / 2**20,
                      "ms_per_submit": 1e3 * (time.time() - t0) / n, "identical": True}), flush=True)
    c.close()


main()
