"""C2 batch time against the number of submits in flight: usage c2_lanes_probe.py SLOTS LANES
(SLOTS streams without the overlap probe, LANES submits in flight from LANES host threads)."""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lamp_amd import capi
if os.environ.get("LGICP_LIB"):   # another build of the library (A/B comparisons)
    capi.LIB_PATH = os.path.abspath(os.environ["LGICP_LIB"])
from lamp_amd.workload import Workload

slots, lanes = int(sys.argv[1]), int(sys.argv[2])
works = [Workload("C2", shard=s, key_offset=100000 * s) for s in range(4)]
cfg = capi.default_config(True, init_method=capi.INIT_IDENTITY)
cfg.reserved[7] = slots
c = capi.Context(cfg)
pks = [[] for _ in range(lanes)]
for w in works:
    w.register(c)
    for f in range(lanes):
        pks[f].append(c.pack_candidates(list(w.pairs)))
print("lanes:", c.set_submit_lanes(lanes), flush=True)


def run(first, n, stride):
    for i in range(first, n, stride):
        c.submit_packed(pks[first][i % 4])


K = 48
run(0, 8, 1)
for rep in range(4):
    th = [threading.Thread(target=run, args=(k, K, lanes)) for k in range(lanes)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    print(f"slots {slots} lanes {lanes} queues {os.environ.get('GPU_MAX_HW_QUEUES')}: {1e3 * (time.perf_counter() - t0) / K:.3f} ms per batch", flush=True)
c.close()
