# FETCH_SIZE / WRITE_SIZE (+ duration) of the kernels matching a regex while a python tool runs: separate --pmc passes,
# counters only (+ --kernel-trace), instrumentation limited to the kernels, every pass under `timeout`.
# usage (GPU box): bash tools/pmc_kernel.sh <name> <kernel regex> <python script + args>
#   writes gpurun_out/pmc_<name>.txt: per kernel name and counter the average per launch (KiB for the two sizes; FETCH_SIZE is
#   doubled by the reader for 16-byte-per-lane reads, /opt/skills/guides/MI355X_MICROARCH.md)
NAME=$1; REGEX=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$NAME.txt
cd /tmp && export TMPDIR=/tmp
: > $OUT
for CTR in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmck_$CTR
  ( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --pmc $CTR --kernel-include-regex "$REGEX" --output-format csv \
      -d /tmp/pmck_$CTR -o p -- python "$@" > /tmp/pmck_$CTR.log 2>&1 )
  echo "== pass $CTR rc=$? : python $@" >> $OUT
  python - $CTR >> $OUT <<'PY'
import csv, glob, collections, sys
c = sys.argv[1]
f = glob.glob('/tmp/pmck_%s/**/*counter_collection.csv' % c, recursive=True)
if not f:
    print('no counter file'); sys.exit(0)
agg = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'][:90]
    agg[k] += float(r['Counter_Value']); n[k] += 1
for k in sorted(agg, key=lambda k: -agg[k]):
    print('%-12s %12.1f per launch (%4d launches)  %s' % (c, agg[k] / n[k], n[k], k))
kt = glob.glob('/tmp/pmck_%s/**/*kernel_trace.csv' % c, recursive=True)
if kt:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(kt[0])):
        d[r['Kernel_Name'][:90]].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
    for k, v in d.items():
        print('%-12s %12.1f us average over %d launches  %s' % ('duration', sum(v) / len(v) / 1e3, len(v), k))
PY
done
cat $OUT
