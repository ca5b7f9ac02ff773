# PMC passes for the wave / LDS / VMEM issue side (SQ counters) of one sparse-conv kernel on the SECOND bs=16 level-L subm
# geometry. Counters only (+ --kernel-trace), separate passes, every pass under `timeout`.
# usage (GPU box): bash tools/pmc_mem_pipeline.sh <level> <fwd|bf16x3|wgrad> [regex]  -> gpurun_out/pmc_mem_<kind>_L<level>.txt
# env for kind bf16x3: CRB_BF16X3_TPW=1|2, CRB_BF16X3_MODE=0..4 (measurement builds)
LEVEL=${1:-3}
KIND=${2:-fwd}
REGEX=${3:-sparse_conv_}
TAG=${KIND}${CRB_BF16X3_MODE:+_mode$CRB_BF16X3_MODE}${CRB_BF16X3_TPW:+_tpw$CRB_BF16X3_TPW}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_mem_${TAG}_L$LEVEL.txt
cd /tmp && export TMPDIR=/tmp
: > $OUT
run_pass () {
  name=$1; shift
  rm -rf /tmp/pmcm_$name
  timeout 240 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$REGEX" --output-format csv \
      -d /tmp/pmcm_$name -o p -- python $GRAFT_REPO_ROOT/tools/pmc_driver.py $LEVEL 3 $KIND > /tmp/pmcm_$name.log 2>&1
  echo "== pass $name rc=$? : $@" >> $OUT
  grep -a PMC_DRIVER /tmp/pmcm_$name.log >> $OUT
  python - $name $REGEX >> $OUT <<'PY'
import csv, glob, collections, sys
f = glob.glob('/tmp/pmcm_%s/**/*counter_collection.csv' % sys.argv[1], recursive=True)
if not f:
    print('no counter file'); sys.exit(0)
agg = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    if sys.argv[2] in r["Kernel_Name"] and 'w_split' not in r["Kernel_Name"]:
        agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for c, x in agg.items():
    print('%-40s %.6g per launch (%d launches)' % (c, x / n[c], n[c]))
kt = glob.glob('/tmp/pmcm_%s/**/*kernel_trace.csv' % sys.argv[1], recursive=True)
if kt:
    d = [float(r['End_Timestamp']) - float(r['Start_Timestamp']) for r in csv.DictReader(open(kt[0]))
         if sys.argv[2] in r['Kernel_Name'] and 'w_split' not in r['Kernel_Name']]
    if d:
        print('%-40s %.1f us average over %d launches (this pass)' % ('kernel duration', sum(d) / len(d) / 1e3, len(d)))
PY
}
# TA_* / TCP_* / TCC_*_sum passes with 6 counters each did not finish inside 240 s per pass on this pool (r02: five passes
# timed out back to back); the per-XCD sums need more hardware counters than one pass offers. Left out; FETCH_SIZE /
# WRITE_SIZE / TCC_HIT_sum / TCC_MISS_sum (two per pass) are collected by tools/pmc_sparse_conv.sh.
run_pass sq GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
run_pass sq2 SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_BUSY_CYCLES
cat $OUT
