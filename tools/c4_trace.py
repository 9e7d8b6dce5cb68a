import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from lamp_amd import capi
from lamp_amd.workload import Workload
work = Workload("C4", pairs=16, keyframes=8)
radius=float(sys.argv[1])
cfg = capi.default_config(True, init_method=capi.INIT_TEASERPP, harris_radius=radius, transform_thresholding=0)
ctx = capi.Context(cfg)
work.register(ctx)
t0=time.perf_counter(); res = ctx.submit(work.candidates()); t1=time.perf_counter()
res = ctx.submit(work.candidates()); t2=time.perf_counter()
print("submit ms", 1e3*(t1-t0), 1e3*(t2-t1))
ctx.close()
