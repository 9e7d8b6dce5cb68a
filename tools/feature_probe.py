"""Times a FEATURES / TEASERPP-initialised submit (GPU box)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lamp_amd import capi, workload

def main():
    kw = dict(a.split("=") for a in sys.argv[1:])
    method = {"features": capi.INIT_FEATURES, "teaser": capi.INIT_TEASERPP}[kw.get("init", "features")]
    n_pairs = int(kw.get("pairs", 64))
    radius = float(kw.get("radius", 0.3))
    wl = workload.Workload("C2", pairs=n_pairs, keyframes=int(kw.get("keyframes", 16)),
                           points=int(kw.get("points", 16384)))
    c = capi.Context(init_method=method, harris_radius=radius, transform_thresholding=0)
    wl.register(c)
    packed = c.pack_candidates(wl.candidates())
    for rep in range(int(kw.get("reps", 3))):
        t = time.time()
        out = c.submit_packed(packed)
        dt = time.time() - t
        st = [out[i].status for i in range(n_pairs)]
        import hashlib
        sig = hashlib.sha1(b"".join(bytes(out[i]) for i in range(n_pairs))).hexdigest()
        print(json.dumps({"rep": rep, "wall_ms": 1e3 * dt, "ms_per_pair": 1e3 * dt / n_pairs,
                          "ok": sum(s == 0 for s in st), "rejected": sum(s == capi.PAIR_INIT_REJECTED for s in st),
                          "total_ms": c.stats().total_ms, "signature": sig}), flush=True)
    c.close()

main()
