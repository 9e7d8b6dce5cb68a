# Kernel times of the feature front end (Harris, FPFH) of cold TEASERPP batches with the chunk boxes
# (default) and without (lgicp_config.reserved[5] bit 2): rocprofv3 --kernel-trace --stats around
# tools/c4_cold_probe.py, once per setting.  usage (GPU box, repository root): bash tools/c4_frontend_kernels.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for R in 0 4; do
  rm -rf gpurun_out/c4k_$R
  PROBE_R5=$R timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4k_$R -o k -- python3 tools/c4_cold_probe.py 128 > gpurun_out/c4k_$R.log 2>&1 || exit 1
  echo "reserved5 $R"
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/c4k_$R/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if any(k in n for k in ("harris","spfh","fpfh","chunk_boxes")):
        print(f'  {n[:45]:45s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
done
