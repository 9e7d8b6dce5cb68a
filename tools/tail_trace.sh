# Kernel trace of the last outer iterations of a lone C2 submit (the batch with the longest straggler
# tail): rocprofv3 --kernel-trace around tools/tail_trace_run.py, then tools/tail_trace.py prints the
# last launches of every stream with their durations and gaps into gpurun_out/tailtrace.txt.
# usage (GPU box, repository root): bash tools/tail_trace.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tailtrace -o tt -- python3 tools/tail_trace_run.py 2 > gpurun_out/tailtrace_run.log 2>&1 && \
F=$(find gpurun_out/tailtrace -name '*kernel_trace.csv' | head -1) && python3 tools/tail_trace.py $F 48 > gpurun_out/tailtrace.txt 2>&1; tail -5 gpurun_out/tailtrace_run.log; head -3 gpurun_out/tailtrace.txt
