"""Where a cold TEASERPP batch goes: three different 512-pair batches one after the other in one
context (the first also pays the context's buffers), then the first again (warm)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from lamp_amd import capi
from lamp_amd.workload import Workload

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_kf = n_pairs // 4
res = [0] * 8
res[5] = int(os.environ.get("PROBE_R5", "0"))   # 4: radius sweeps of the front end without chunk boxes
c = capi.Context(init_method=capi.INIT_TEASERPP, harris_radius=0.5, transform_thresholding=0,
                 reserved=(capi.C.c_int * 8)(*res))
works = [Workload("C4", pairs=n_pairs, keyframes=n_kf, shard=s, key_offset=100000 * s) for s in range(3)]
for i, w in enumerate(works + works[:1]):
    t0 = time.perf_counter()
    if i < 3:
        w.register(c)
    t1 = time.perf_counter()
    sys.stderr.write(f"==== batch {i} ====\n")
    c.submit(w.candidates())
    t2 = time.perf_counter()
    print(f"batch {i}: register {1e3 * (t1 - t0):.1f} ms, submit {1e3 * (t2 - t1):.1f} ms", flush=True)
c.close()
