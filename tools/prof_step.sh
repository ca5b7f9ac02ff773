cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o sec -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --scoring-pool 0 --pvrcnn-steps 0 --bf16x3-steps 0 --miopen-steps 0 > gpurun_out/${R:-r06}_bench_under_rocprof_$V.json 2>gpurun_out/prof_err.log
PROF_LIST=chain_,table_rows,tables_,hash_build,fillBuffer,vox_,nbr_permute PROF_SPLIT_GRID=sparse_conv_fwd2 PROF_GAPS=12 python tools/prof_summary.py $(find /tmp/prof -name "*kernel_trace.csv" | head -1) 8 > gpurun_out/${R:-r06}_second_bs16_steady_state_kernel_summary_$V.csv
head -1 gpurun_out/${R:-r06}_second_bs16_steady_state_kernel_summary_$V.csv
