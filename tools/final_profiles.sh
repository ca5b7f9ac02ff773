# round-end evidence on the GPU box: PMC passes of the Winograd kernel, step / PV-RCNN / scoring kernel traces.  usage: bash tools/final_profiles.sh <version tag>
V=$1
cd $GRAFT_REPO_ROOT
bash tools/pmc_sq.sh wino2_$V winograd2_kernel tools/pmc_wino2.py > /dev/null 2>&1
bash tools/prof_step.sh $V
bash tools/prof_pvrcnn.sh $V
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/psc; rocprofv3 --kernel-trace --output-format csv -d /tmp/psc -o sc -- python tools/prof_scoring_resident.py 64 > gpurun_out/prof_sc.log 2>&1
PROF_MARKER=vox_insert PROF_GAPS=12 python tools/prof_summary.py $(find /tmp/psc -name "*kernel_trace.csv" | head -1) 6 > gpurun_out/r04_crb_scoring_bs64_kernel_summary_$V.csv
head -1 gpurun_out/r04_crb_scoring_bs64_kernel_summary_$V.csv
grep "FETCH_SIZE\|WRITE_SIZE" gpurun_out/pmc_sq_wino2_$V.txt
