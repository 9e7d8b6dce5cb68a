"""C2 throughput with two submits in flight on one context (lgicp_set_submit_lanes(2), two host
threads: one submit's latency-bound last iterations run beside the other's full-size launches)
against one submit at a time."""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lamp_amd import capi
from lamp_amd.workload import Workload

works = [Workload("C2", shard=s, key_offset=100000 * s) for s in range(4)]
c = capi.Context(capi.default_config(True, init_method=capi.INIT_IDENTITY))
pks = []
for w in works:
    w.register(c)
    pks.append(c.pack_candidates(list(w.pairs)))


def run(first, n, stride):
    for i in range(first, n, stride):
        c.submit_packed(pks[i % 4])


K = 24
run(0, 8, 1)
for rep in range(2):
    t0 = time.perf_counter()
    run(0, K, 1)
    print(f"one submit at a time: {1e3 * (time.perf_counter() - t0) / K:.3f} ms per batch", flush=True)
print("lanes:", c.set_submit_lanes(2), flush=True)
run(0, 8, 1)
for rep in range(2):
    t0 = time.perf_counter()
    run(0, K, 1)
    print(f"two lanes, one caller: {1e3 * (time.perf_counter() - t0) / K:.3f} ms per batch", flush=True)
def threaded(n_threads, what):
    for rep in range(3):
        th = [threading.Thread(target=run, args=(k, K, n_threads)) for k in range(n_threads)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        print(f"{what}: {1e3 * (time.perf_counter() - t0) / K:.3f} ms per batch", flush=True)


threaded(2, "two lanes, two callers")
threaded(3, "two lanes, three callers")
print("lanes:", c.set_submit_lanes(4), flush=True)
run(0, 8, 1)
threaded(3, "four lanes, three callers")
threaded(4, "four lanes, four callers")
c.close()
