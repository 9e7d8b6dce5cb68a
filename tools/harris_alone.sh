# Durations of the feature front end's kernels (Harris, FPFH) when one keyframe is processed alone:
# rocprofv3 --kernel-trace --stats around tools/harris_alone_probe.py.
# usage (GPU box, repository root): bash tools/harris_alone.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ha
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ha -o k -- python3 tools/harris_alone_probe.py > gpurun_out/ha.log 2>&1 || exit 1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/ha/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if any(k in n for k in ("harris","spfh","fpfh","chunk_boxes")):
        print(f'  {n[:45]:45s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us min {float(r["MinNs"])/1e3:8.1f}')
PY
