# SQ / LDS / TCP counters (+ duration) of the kernels matching a regex while a python tool runs: separate --pmc passes of <= 8 SQ
# counters, counters only (+ --kernel-trace), every pass under `timeout`.
# usage (GPU box): bash tools/pmc_sq.sh <name> <kernel regex> <python script + args>   -> gpurun_out/pmc_sq_<name>.txt
NAME=$1; REGEX=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq_$NAME.txt
cd /tmp && export TMPDIR=/tmp
: > $OUT
P=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU SQ_INSTS_MFMA" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_ACTIVE_INST_SCA" \
            "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  P=$((P+1))
  rm -rf /tmp/pmcs_$P
  ( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --kernel-include-regex "$REGEX" --output-format csv \
      -d /tmp/pmcs_$P -o p -- python "$@" > /tmp/pmcs_$P.log 2>&1 )
  echo "== pass $P rc=$? ($CTRS): python $@" >> $OUT
  python - $P >> $OUT <<'PY'
import csv, glob, collections, sys
p = sys.argv[1]
f = glob.glob('/tmp/pmcs_%s/**/*counter_collection.csv' % p, recursive=True)
if not f:
    print('no counter file'); print(open('/tmp/pmcs_%s.log' % p).read()[-1500:]); sys.exit(0)
agg = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = (r['Kernel_Name'][:60], r['Counter_Name'])
    agg[k] += float(r['Counter_Value']); n[k] += 1
for k in sorted(agg):
    print('%-36s %16.1f per launch (%3d launches)  %s' % (k[1], agg[k] / n[k], n[k], k[0]))
kt = glob.glob('/tmp/pmcs_%s/**/*kernel_trace.csv' % p, recursive=True)
if kt:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(kt[0])):
        d[r['Kernel_Name'][:60]].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
    for k, v in d.items():
        print('%-36s %16.1f us average over %d launches  %s' % ('duration', sum(v) / len(v) / 1e3, len(v), k))
PY
done
cat $OUT
