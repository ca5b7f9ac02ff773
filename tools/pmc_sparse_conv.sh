cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc1 -o p -- python $GRAFT_REPO_ROOT/tools/bench_sparse_conv.py --levels 3 --iters 3 > /tmp/pmc1.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmc1/*counter_collection.csv')
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r['Kernel_Name'][:70]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in agg.items():
    if 'sparse_conv' in k:
        print(k, {c: '%.3g' % x for c, x in v.items()})
PY
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pmc2 -o p -- python $GRAFT_REPO_ROOT/tools/bench_sparse_conv.py --levels 3 --iters 3 > /tmp/pmc2.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmc2/*counter_collection.csv')
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r['Kernel_Name'][:70]
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k, v in agg.items():
    if 'sparse_conv' in k:
        print(k, {c: '%.4g per launch' % (x / n[(k, c)]) for c, x in v.items()})
PY
