cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tailtrace -o tt -- python3 tools/tail_trace_run.py 2 > gpurun_out/tailtrace_run.log 2>&1 && \
F=$(find gpurun_out/tailtrace -name '*kernel_trace.csv' | head -1) && python3 tools/tail_trace.py $F 48 > gpurun_out/tailtrace.txt 2>&1; tail -5 gpurun_out/tailtrace_run.log; head -3 gpurun_out/tailtrace.txt
