"""The lookup certificates must not change a single bit of a submit's results: runs one workload
with certificates (default) and without (reserved[6] = 2) and compares every result field."""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd import capi
from lamp_amd.workload import Workload


def run(work, mode, steps):
    cfg = capi.default_config(True, device=0, init_method=capi.INIT_IDENTITY)
    cfg.reserved[6] = mode
    ctx = capi.Context(cfg)
    work.register(ctx)
    packed = ctx.pack_candidates(work.candidates())
    n = len(work.pairs)
    for _ in range(3):
        out = ctx.submit_packed(packed)
    s0 = ctx.stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = ctx.submit_packed(packed)
    dt = (time.perf_counter() - t0) / steps
    s1 = ctx.stats()
    rec = []
    for i in range(n):
        r = out[i]
        rec.append((r.status, r.iterations, r.n_correspondences, r.converged, bytes(np.array(r.T_icp, np.float32)),
                    np.float64(r.fitness).tobytes(), bytes(np.array(r.covariance, np.float64)),
                    bytes(np.array(r.information, np.float64))))
    ev = s1.nn_executed_evals
    ctx.close()
    return rec, dt * 1e3, ev


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "C2"
    pairs = int(sys.argv[2]) if len(sys.argv) > 2 else None
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    work = Workload(config, pairs=pairs)
    a, ms_a, ev_a = run(work, 0, steps)
    b, ms_b, ev_b = run(work, 2, steps)
    diff = [i for i in range(len(a)) if a[i] != b[i]]
    sig = hashlib.sha1(b"".join(b"".join(x if isinstance(x, bytes) else str(x).encode() for x in r) for r in a)).hexdigest()
    print(json.dumps({"config": config, "pairs": len(a), "differing_pairs": len(diff), "first": diff[:5], "signature": sig,
                      "ms_with_certificates": ms_a, "ms_without": ms_b,
                      "executed_evals_with": ev_a, "executed_evals_without": ev_b}), flush=True)
    sys.exit(1 if diff else 0)


main()
