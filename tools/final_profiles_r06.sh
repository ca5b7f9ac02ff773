# round-6 step / scoring profiles of the final code (GPU box): SECOND step, PV-RCNN step, 16- and 64-frame scoring passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=${1:-v3}
bash tools/prof_step.sh $V
cp /tmp/prof/*/*_kernel_stats.csv gpurun_out/${R:-r06}_bench_kernel_stats_$V.csv 2>/dev/null || find /tmp/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/${R:-r06}_bench_kernel_stats_$V.csv \;
PROF_LIST=winograd4_kernel,winograd2_wgrad,fps2_kernel,sa_train_kernel bash tools/prof_pvrcnn.sh $V
for B in 16 64; do
  rm -rf /tmp/profsc
  rocprofv3 --kernel-trace --output-format csv -d /tmp/profsc -o sc -- python tools/prof_scoring_resident.py $B > /tmp/sc.log 2>&1
  PROF_MARKER=vox_insert PROF_GAPS=6 python tools/prof_summary.py $(find /tmp/profsc -name "*kernel_trace.csv" | head -1) 6 > gpurun_out/${R:-r06}_crb_scoring_bs${B}_kernel_summary_$V.csv
  head -1 gpurun_out/${R:-r06}_crb_scoring_bs${B}_kernel_summary_$V.csv
done
