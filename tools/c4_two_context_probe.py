"""Would TEASER++-initialised submits gain from running side by side?  N contexts on one GPU,
each with its own C4 batch, submitting from N threads, against the same submits in turn.
usage: c4_two_context_probe.py [PAIRS [N]]"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lamp_amd import capi
from lamp_amd.workload import Workload

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2
works = [Workload("C4", pairs=n_pairs, shard=s, key_offset=100000 * s) for s in range(n_ctx)]


def make(w):
    cfg = capi.default_config(True, init_method=capi.INIT_TEASERPP)
    cfg.harris_radius = 0.5
    c = capi.Context(cfg)
    w.register(c)
    return c, c.pack_candidates(list(w.pairs))


ctxs = [make(w) for w in works]
for c, pk in ctxs:
    c.submit_packed(pk)          # features computed, buffers sized
K = 2
t0 = time.perf_counter()
for c, pk in ctxs:
    for _ in range(K):
        c.submit_packed(pk)
dt = time.perf_counter() - t0
print(f"in turn: {1e3 * dt / (n_ctx * K):.1f} ms per batch of {n_pairs}", flush=True)


def run(c, pk):
    for _ in range(K):
        c.submit_packed(pk)


th = [threading.Thread(target=run, args=cp) for cp in ctxs]
t0 = time.perf_counter()
for t in th:
    t.start()
for t in th:
    t.join()
dt = time.perf_counter() - t0
print(f"{n_ctx} side by side: {1e3 * dt / (n_ctx * K):.1f} ms per batch of {n_pairs}", flush=True)
for c, _ in ctxs:
    c.close()
