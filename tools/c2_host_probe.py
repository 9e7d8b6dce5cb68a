"""Wall-clock phases of a C2 submit seen from the caller and from the library's trace marks."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lamp_amd import capi
from lamp_amd.workload import Workload

work = Workload("C2")
c = capi.Context(capi.default_config(True, init_method=capi.INIT_IDENTITY))
work.register(c)
pk = c.pack_candidates(list(work.pairs))
for _ in range(6):
    c.submit_packed(pk)
sys.stderr.write("==== timed submit ====\n")
t0 = time.perf_counter()
c.submit_packed(pk)
t1 = time.perf_counter()
sys.stderr.write("submit %.3f ms\n" % (1e3 * (t1 - t0)))
c.close()
