# FETCH_SIZE of every launch of the LDS-DMA calibration probe, in launch order (calibrate the counter on the
# Winograd forward kernel's access pattern), then the same counter for the Winograd forward kernel itself.
# usage (GPU box): bash tools/pmc_lds_dma_calib.sh   -> gpurun_out/r05_pmc_lds_dma_calib.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_pmc_lds_dma_calib.txt
cd /tmp && export TMPDIR=/tmp
: > $OUT
rm -rf /tmp/pmccal
( cd $GRAFT_REPO_ROOT && CRB_MEASURE_LIB=1 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "probe_lds_dma" \
    --output-format csv -d /tmp/pmccal -o p -- python tools/pmc_lds_dma_calib.py > /tmp/pmccal.log 2>&1 )
echo "== FETCH_SIZE pass rc=$? : python tools/pmc_lds_dma_calib.py" >> $OUT
grep PATTERN /tmp/pmccal.log >> $OUT
python - >> $OUT <<'PY'
import csv, glob
f = glob.glob('/tmp/pmccal/**/*counter_collection.csv', recursive=True)
if not f:
    print('no counter file'); print(open('/tmp/pmccal.log').read()[-1500:])
else:
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Dispatch_Id']))
    vals = [float(r['Counter_Value']) for r in rows if r['Counter_Name'] == 'FETCH_SIZE']
    for k in range(0, len(vals), 3):
        v = vals[k:k + 3]
        print('pattern %d: FETCH_SIZE per launch (KiB) %s -> %.1f MB' % (k // 3, ['%.0f' % x for x in v], sum(v) / len(v) * 1024 / 1e6))
PY
for SHAPE in "16 128 128 200 176" "16 256 256 100 88" "16 256 128 200 176" "16 128 256 200 176"; do
for CTR in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcw_$CTR
  ( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --pmc $CTR --kernel-include-regex "winograd2_kernel" --output-format csv \
      -d /tmp/pmcw_$CTR -o p -- python tools/pmc_wino2.py $SHAPE > /tmp/pmcw_$CTR.log 2>&1 )
  echo "== winograd2_kernel N C K H W = $SHAPE, $CTR pass rc=$?" >> $OUT
  python - $CTR >> $OUT <<'PY'
import csv, glob, sys
c = sys.argv[1]
f = glob.glob('/tmp/pmcw_%s/**/*counter_collection.csv' % c, recursive=True)
if not f:
    print('no counter file')
else:
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(f[0])) if r['Counter_Name'] == c and 'winograd2_kernel' in r['Kernel_Name']]
    print('%s per launch (KiB): %s' % (c, ['%.0f' % v for v in vals]))
PY
done
done
cat $OUT
