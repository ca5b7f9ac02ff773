# PMC passes for the memory pipeline (TA / TCP / TCC / LDS) of one sparse-conv kernel on the SECOND bs=16 level-L subm
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
run_pass ta TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run_pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run_pass tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum
run_pass tcc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_sum
run_pass tcc2 TCC_CYCLE_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_READ_SECTORS_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum
run_pass sq GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
run_pass sq2 SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_BUSY_CYCLES
cat $OUT
