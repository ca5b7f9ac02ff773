cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profpv -o pv -- python tools/bench_pvrcnn.py --steps 10 --warmup 3 > gpurun_out/${R:-r06}_pvrcnn_under_rocprof_$V.json 2>gpurun_out/prof_err.log
PROF_LIST=${PROF_LIST:-} PROF_GAPS=12 python tools/prof_summary.py $(find /tmp/profpv -name "*kernel_trace.csv" | head -1) 6 > gpurun_out/${R:-r06}_pvrcnn_bs16_steady_state_kernel_summary_$V.csv
head -1 gpurun_out/${R:-r06}_pvrcnn_bs16_steady_state_kernel_summary_$V.csv
