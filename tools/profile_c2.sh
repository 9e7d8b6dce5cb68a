set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01b
rm -rf $O; mkdir -p $O
timeout -k 10 280 python3 bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 10 --warmup 2 > $O/bench_rocprof.json 2> $O/kt.err
tail -1 $O/bench_rocprof.json
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p1 -o p1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu --slots 4 > $O/p1.json 2> $O/p1.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p3 -o p3 -- python3 bench.py --steps 3 --warmup 1 --no-cpu --slots 4 > $O/p3.json 2> $O/p3.err
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/p2 -o p2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu --slots 4 > $O/p2.json 2> $O/p2.err
python3 tools/pmc_summary.py $O/pmc_summary.json "C2 (256 pairs x 16384 points), 3 timed steps + 1 warm-up" "rocprofv3 --pmc <one counter group per run> -- python3 bench.py --steps 3 --warmup 1 --no-cpu --slots 4" $O/p1/*counter_collection.csv $O/p3/*counter_collection.csv $O/p2/*counter_collection.csv
ls $O
