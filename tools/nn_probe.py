"""Prints per-stage times and the executed/dense distance ratio of one C2 submit (GPU box)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd import capi, workload

def main():
    kw = {}
    for a in sys.argv[1:]:
        k, v = a.split("=")
        kw[k] = int(v)
    with_torch = kw.pop("torch", 0)   # 1: import torch and touch the GPU first; 2: also a process group
    if with_torch:
        import torch
        torch.cuda.set_device(0)
        torch.zeros(1, device="cuda")
        if with_torch >= 2:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
            if with_torch >= 3:
                t = torch.zeros(8, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
    if with_torch:
        print("hip runtimes:", sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "librccl" in l or "libhsa-runtime" in l}), flush=True)
    res = [0] * 8
    res[2] = kw.pop("exhaustive", 0)
    res[3] = kw.pop("r0mm", 0)
    res[4] = kw.pop("capmm", 0)
    res[7] = kw.pop("slots", 0)
    iters = kw.pop("iters", 200)
    n_pairs = kw.pop("pairs", 256)
    inner = kw.pop("inner", 50)
    wl = workload.Workload("C2", pairs=n_pairs)
    c = capi.Context(init_method=capi.INIT_IDENTITY, iterations=iters, optimizer_iterations=inner, stage_timing=1, reserved=(capi.C.c_int * 8)(*res))
    wl.register(c)
    packed = c.pack_candidates(wl.candidates())
    reps = kw.pop("reps", 4)
    for rep in range(reps):
        t = time.time()
        out = c.submit_packed(packed)
        dt = time.time() - t
        out = [out[i] for i in range(packed[1])]
        s = c.stats()
        print(json.dumps({"rep": rep, "wall_ms": 1e3 * dt, "nn_ms": s.nn_ms, "reduce_ms": s.reduce_ms,
                          "solve_ms": s.solve_ms, "cov_ms": s.cov_ms, "total_ms": s.total_ms,
                          "final_nn_ms": s.last_nn_kernel_ms,
                          "dense_evals_per_pass": s.last_nn_evals_per_launch,
                          "executed_evals": s.nn_executed_evals,
                          "ok": sum(r.status == 0 for r in out)}))
    c.close()

main()
