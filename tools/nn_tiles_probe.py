"""Tiles tested against tiles walked by the pruned 1-NN kernel on a C2 batch: run with the
regular library and with a build of knn_pruned.hip that defines NNP_COUNT_CONSIDERED
(LGICP_LIB=<path>)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lamp_amd import capi
if os.environ.get("LGICP_LIB"):
    capi.LIB_PATH = os.environ["LGICP_LIB"]
from lamp_amd.workload import Workload

work = Workload("C2")
cfg = capi.default_config(True, init_method=capi.INIT_IDENTITY, stage_timing=1)
cfg.reserved[7] = 1
c = capi.Context(cfg)
work.register(c)
for _ in range(2):
    c.submit(work.candidates())
s = c.stats()
print("tile units (x 32 x 64 distances):", s.nn_executed_evals / (32 * 64), "nn_ms", s.nn_ms, "launches", s.nn_launches)
c.close()
