"""Submits the same C2 batch many times: results must not drift, host and device memory must not grow.
usage: soak.py [submits] [plain|accumulated|accumulated-tight|bfgs]   (accumulated: consecutive keys,
targets of five scans kept on the device; -tight: a budget of a few clouds, so that entries are
evicted and rebuilt in every submit; bfgs: the reference's solver on the device)"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import psutil
from lamp_amd import capi, workload


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    mode = sys.argv[2] if len(sys.argv) > 2 else "plain"
    wl = workload.Workload("C2", consecutive_keys=mode.startswith("accumulated"))
    c = capi.Context(init_method=capi.INIT_IDENTITY,
                     accumulated_cache_bytes=(96 << 20) if mode == "accumulated-tight" else 0,
                     solver=capi.SOLVER_BFGS if mode == "bfgs" else capi.SOLVER_GAUSS_NEWTON)
    hip = ctypes.CDLL("libamdhip64.so.7")

    def device_free_mb():
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
        return free.value / 2**20

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
        if k == 20:
            dev0 = device_free_mb()
        elif not np.array_equal(T, first):
            print(json.dumps({"submit": k, "error": "results changed"}), flush=True)
            sys.exit(1)
        if k % 250 == 0:
            print(json.dumps({"submit": k, "rss_mb": proc.memory_info().rss / 2**20,
                              "device_free_mb": device_free_mb(), "elapsed_s": time.time() - t0}), flush=True)
    print(json.dumps({"submits": n, "mode": mode, "rss_growth_mb": (proc.memory_info().rss - rss0) / 2**20,
                      "device_free_change_mb_since_submit_20": device_free_mb() - dev0,
                      "ms_per_submit": 1e3 * (time.time() - t0) / n, "identical": True}), flush=True)
    c.close()


main()
