"""Does the C2 step time depend on how long the GPU has been busy?  Four lanes, four callers,
the time of every block of 24 batches over a few seconds."""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lamp_amd import capi
from lamp_amd.workload import Workload

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
works = [Workload("C2", shard=s, key_offset=100000 * s) for s in range(4)]
c = capi.Context(capi.default_config(True, init_method=capi.INIT_IDENTITY))
pks = [[], [], [], []]
for w in works:
    w.register(c)
    for f in range(4):
        pks[f].append(c.pack_candidates(list(w.pairs)))
c.set_submit_lanes(lanes)


def run(first, n, stride):
    for i in range(first, n, stride):
        c.submit_packed(pks[first][i % 4])


K = 24
run(0, 8, 1)
for rep in range(40):
    th = [threading.Thread(target=run, args=(k, K, lanes)) for k in range(lanes)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    print(f"block {rep}: {1e3 * (time.perf_counter() - t0) / K:.3f} ms per batch", flush=True)
c.close()
