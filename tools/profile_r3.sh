# Round-3 evidence for profiles/: the default bench line, a kernel trace and the PMC passes of the
# one-stream run (kernel durations exclusive, sum of kernel time <= step time).
# usage (on the GPU box, from the repository root): bash tools/profile_r3.sh [CONFIG]
set -e
CFG=${1:-C2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_$CFG
rm -rf $O; mkdir -p $O
STEPS=8; WARM=4; FULL=""; PT=120
if [ "$CFG" = "C4" ]; then STEPS=2; WARM=1; FULL="--steps 6 --warmup 2 --cpu-sample 16"; PT=280; fi
ONE="--config $CFG --slots 1 --steps $STEPS --warmup $WARM --no-cpu --no-exclusive-phase --no-cold --no-sub-records"
timeout -k 10 580 python3 bench.py --config $CFG $FULL > $O/bench.json 2> $O/bench.err
timeout -k 10 280 python3 bench.py $ONE > $O/bench_one_stream.json 2> $O/bench_one_stream.err
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py $ONE > $O/bench_one_stream_rocprof.json 2> $O/kt.err
timeout -s KILL $PT rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p1 -o p1 -- python3 bench.py $ONE > $O/p1.json 2> $O/p1.err
timeout -s KILL $PT rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p3 -o p3 -- python3 bench.py $ONE > $O/p3.json 2> $O/p3.err
timeout -s KILL $PT rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -o p2 -- python3 bench.py $ONE > $O/p2.json 2> $O/p2.err
python3 tools/pmc_summary.py $O/pmc_summary.json "$CFG, one stream, $STEPS timed steps + $WARM warm-up" "rocprofv3 --pmc <one counter group per run> -- python3 bench.py $ONE" $O/p1/*counter_collection.csv $O/p3/*counter_collection.csv $O/p2/*counter_collection.csv
ls $O $O/kt
