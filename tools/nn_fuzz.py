"""Random clouds of many shapes through both 1-NN kernels (pruned and exhaustive): every query must
get the same (index, distance bits) from both, and both must agree with a float32 numpy sweep."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd import capi


def make(rng, n, kind):
    if kind == 0:      # uniform box
        x = rng.uniform(-20, 20, size=(n, 3))
    elif kind == 1:    # a few tight clusters
        c = rng.uniform(-30, 30, size=(max(1, n // 200 + 1), 3))
        x = c[rng.integers(0, len(c), n)] + rng.normal(scale=0.05, size=(n, 3))
    elif kind == 2:    # planes (surfaces, like scans)
        x = rng.uniform(-10, 10, size=(n, 3))
        x[:, rng.integers(0, 3)] = np.round(x[:, 0])
    elif kind == 3:    # lattice: many exact ties
        x = rng.integers(-6, 6, size=(n, 3)).astype(np.float64) * 0.5
    else:              # line
        x = np.outer(rng.uniform(-50, 50, n), rng.normal(size=3))
    return x.astype(np.float32)


def brute(src, tgt):
    idx = np.empty(len(src), np.int32)
    d2 = np.empty(len(src), np.float32)
    for s in range(0, len(src), 256):
        q = src[s:s + 256]
        dx = q[:, None, 0] - tgt[None, :, 0]
        dy = q[:, None, 1] - tgt[None, :, 1]
        dz = q[:, None, 2] - tgt[None, :, 2]
        d = (dx * dx + dy * dy) + dz * dz            # float32, same operation order
        d = np.where(np.isnan(d), np.inf, d)
        k = d.argmin(1)                               # first minimum = lowest index
        idx[s:s + 256], d2[s:s + 256] = k, d[np.arange(len(q)), k]
    return idx, d2


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    rng = np.random.default_rng(20261111)
    res = [0] * 8
    ctx = {}
    for ex in (0, 1):
        res[2] = ex
        ctx[ex] = capi.Context(reserved=(capi.C.c_int * 8)(*res))
    bad = 0
    for case in range(cases):
        n_src = int(rng.choice([1, 2, 31, 64, 65, 500, 3000, 7000]))
        n_tgt = int(rng.choice([1, 2, 31, 32, 33, 100, 1024, 1025, 5000, 9000]))
        src, tgt = make(rng, n_src, int(rng.integers(0, 5))), make(rng, n_tgt, int(rng.integers(0, 5)))
        if rng.random() < 0.3:
            tgt[rng.integers(0, n_tgt, max(1, n_tgt // 10))] = tgt[0]      # duplicates
        if rng.random() < 0.2:
            src[rng.integers(0, n_src)] = np.nan
        if rng.random() < 0.2 and n_tgt > 4:
            tgt[rng.integers(1, n_tgt)] = np.inf
        T = np.eye(4, dtype=np.float32)
        T[:3, 3] = rng.normal(scale=2.0, size=3)
        a = ctx[0].nn_search(T, src, tgt)
        b = ctx[1].nn_search(T, src, tgt)
        q = (src @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
        ok = np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
        if T[:3, :3].tolist() == np.eye(3).tolist():                        # pure translation: numpy reproduces q
            fin = np.isfinite(q).all(1)
            ri, rd = brute(q[fin], tgt)
            found = np.isfinite(rd)
            ok = ok and np.array_equal(a[0][fin][found], ri[found]) and \
                np.array_equal(a[1][fin][found].view(np.uint32), rd[found].view(np.uint32))
        if not ok:
            bad += 1
            print(json.dumps({"case": case, "n_src": n_src, "n_tgt": n_tgt, "mismatch": True}), flush=True)
    for c in ctx.values():
        c.close()
    print(json.dumps({"cases": cases, "mismatches": bad}), flush=True)
    sys.exit(1 if bad else 0)


main()
