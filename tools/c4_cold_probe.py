"""Times the feature front end of a cold TEASERPP submit (keyframes never seen before)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from lamp_amd import capi
from lamp_amd.workload import Workload

work = Workload("C4", pairs=64, keyframes=32)
c = capi.Context(init_method=capi.INIT_TEASERPP, harris_radius=0.5, transform_thresholding=0)
work.register(c)
t0 = time.perf_counter()
c.submit(work.candidates())
t1 = time.perf_counter()
c.submit(work.candidates())
t2 = time.perf_counter()
print("cold submit ms", 1e3 * (t1 - t0), "warm", 1e3 * (t2 - t1))
c.close()
