"""Reads a rocprofv3 kernel trace (csv) and prints, for the busiest second half of it, how the wall
time splits by what was running: lookups, reductions, only small kernels, nothing."""
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    k = "nn" if "nn_pruned" in n else "reduce" if "gicp_reduce" in n else "solve" if "gicp_solve" in n else "other"
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
t0 = min(e[0] for e in ev); t1 = max(e[1] for e in ev)
lo = t0 + (t1 - t0) // 2
pts = []
for s, e, k in ev:
    if e <= lo:
        continue
    s = max(s, lo)
    pts.append((s, 1, k)); pts.append((e, -1, k))
pts.sort()
cnt = dict(nn=0, reduce=0, solve=0, other=0)
acc = {}
prev = lo
for t, d, k in pts:
    if t > prev:
        key = ("nn+reduce" if cnt["nn"] and cnt["reduce"] else "nn" if cnt["nn"] else "reduce" if cnt["reduce"]
               else "solve/other only" if cnt["solve"] or cnt["other"] else "idle")
        acc[key] = acc.get(key, 0) + (t - prev)
        prev = t
    cnt[k] += d
tot = sum(acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{k:18s} {100.0 * v / tot:5.1f} %")
busy = {k: 0 for k in cnt}
for s, e, k in ev:
    if e > lo:
        busy[k] += e - max(s, lo)
print("kernel time / wall:", {k: round(v / tot, 3) for k, v in busy.items()})
