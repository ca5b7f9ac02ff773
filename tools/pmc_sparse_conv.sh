# PMC passes for the subm gather-GEMM forward (level-3 geometry of SECOND bs=16). Separate passes (SQ / FETCH_SIZE /
# WRITE_SIZE), counters only (+ --kernel-trace), instrumentation limited to the conv kernel, every pass under `timeout`.
# usage (GPU box): bash tools/pmc_sparse_conv.sh [level] [fwd|wgrad|bf16x3] [kernel regex] ; writes gpurun_out/pmc_sparse_conv_<kind>_L<level>.txt
LEVEL=${1:-3}
KIND=${2:-fwd}
REGEX=${3:-sparse_conv_$KIND}      # kind bf16x3: pass sparse_conv_fwd_bf16x3 (env CRB_BF16X3_TPW / _MODE as in tools/pmc_driver.py)
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sparse_conv_${KIND}_L$LEVEL.txt
cd /tmp && export TMPDIR=/tmp
: > $OUT
run_pass () {   # name, counters...
  name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 240 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$REGEX" --output-format csv \
      -d /tmp/pmc_$name -o p -- python $GRAFT_REPO_ROOT/tools/pmc_driver.py $LEVEL 3 $KIND > /tmp/pmc_$name.log 2>&1
  echo "== pass $name rc=$? : $@" >> $OUT
  grep -a PMC_DRIVER /tmp/pmc_$name.log >> $OUT
  python - $name $REGEX >> $OUT <<'PY'
import csv, glob, collections, sys
f = glob.glob('/tmp/pmc_%s/**/*counter_collection.csv' % sys.argv[1], recursive=True)
if not f:
    print('no counter file'); sys.exit(0)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(float); n = collections.Counter()
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for c, x in agg.items():
    print('%-28s %.6g per launch (%d launches)' % (c, x / n[c], n[c]))
kt = glob.glob('/tmp/pmc_%s/**/*kernel_trace.csv' % sys.argv[1], recursive=True)
if kt:
    d = [float(r['End_Timestamp']) - float(r['Start_Timestamp']) for r in csv.DictReader(open(kt[0])) if sys.argv[2] in r['Kernel_Name']]
    if d:
        print('%-28s %.1f us average over %d launches (this pass)' % ('kernel duration', sum(d) / len(d) / 1e3, len(d)))
PY
}
run_pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS
run_pass sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run_pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
run_pass tcc TCC_HIT_sum TCC_MISS_sum
cat $OUT
