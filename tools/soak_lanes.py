"""Soak for submits side by side: LANES host threads submit rotating C2 batches N times each on one
context; every result is compared, bit for bit, with what the same batch returned when it was
submitted alone.  SOAK_REGISTER=1 adds a thread that keeps registering new keyframes and verifying
them meanwhile.  usage: soak_lanes.py [LANES [N]]"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from lamp_amd import capi
from lamp_amd.workload import Workload

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n_rep = int(sys.argv[2]) if len(sys.argv) > 2 else 200
works = [Workload("C2", shard=s, key_offset=100000 * s) for s in range(4)]
c = capi.Context(capi.default_config(True, init_method=capi.INIT_IDENTITY))
for w in works:
    w.register(c)


def signature(out, n):
    return b"".join(bytes(memoryview(out[i]).cast("B")) for i in range(n))


ref = []
for w in works:
    pk = c.pack_candidates(list(w.pairs))
    ref.append(signature(c.submit_packed(pk), len(w.pairs)))
print("lanes:", c.set_submit_lanes(lanes), flush=True)
bad, done = [], [0] * lanes


def run(k):
    pks = [c.pack_candidates(list(w.pairs)) for w in works]
    for i in range(n_rep):
        b = (i + k) % 4
        out = c.submit_packed(pks[b])
        if signature(out, len(works[b].pairs)) != ref[b]:
            bad.append((k, i, b))
        done[k] += 1


n_reg = [0]


def registrar():
    """new keyframes every few milliseconds while the lanes are busy, each batch of them verified
    by a submit of its own (which competes for a lane like any other)"""
    s = 0
    while any(d < n_rep for d in done):
        w = Workload("C2", pairs=8, points=2048, keyframes=4, shard=100 + s, key_offset=10000000 + 1000 * s)
        w.register(c)
        res = c.submit(w.candidates())
        if sum(r.status == 0 for r in res) < 4:
            bad.append(("late", s, [r.status for r in res]))
        n_reg[0] += 1
        s += 1
        time.sleep(0.003)


t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(k,)) for k in range(lanes)]
if os.environ.get("SOAK_REGISTER"):
    th.append(threading.Thread(target=registrar))
for t in th:
    t.start()
while any(t.is_alive() for t in th):
    time.sleep(5)
    print(f"{sum(done)} submits, {n_reg[0]} registrations, {len(bad)} mismatches", flush=True)
for t in th:
    t.join()
dt = time.perf_counter() - t0
print(f"{sum(done)} submits in {dt:.1f} s ({1e3 * dt / sum(done):.3f} ms per batch), mismatches: {bad[:5]}")
c.close()
sys.exit(1 if bad else 0)
