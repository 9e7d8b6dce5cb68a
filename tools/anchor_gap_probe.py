#!/usr/bin/env python3
"""Device (lgicp_submit, default solver) against the numpy + MINPACK implementation of
tests/test_independent_anchors.py on one C2 and one C3 pair: prints the pose gap and whether the float
matrices are identical.  usage (GPU box, repository root): python tools/anchor_gap_probe.py"""
import sys; sys.path.insert(0,'.')
import numpy as np
from oracle import oracle as O
from lamp_amd import capi
from lamp_amd.workload import Workload
from tests.test_independent_anchors import _gicp_numpy
for config in ("C2","C3"):
    work = Workload(config, pairs=1, keyframes=2)
    (a,b),=work.pairs
    src,sn=work.keyframes[a]; tgt,tn=work.keyframes[b]
    T_np=_gicp_numpy(src,sn,tgt,tn,iterations=60)
    c=capi.Context(init_method=capi.INIT_IDENTITY, transform_thresholding=0)
    work.register(c); d=capi.result_to_dict(c.submit(work.candidates())[0]); c.close()
    print(config, O.pose_gap(T_np,d["T_icp"]), "identical", np.array_equal(T_np, d["T_icp"]), "iterations", d["iterations"])
