#!/bin/bash
# round 3, GPU session aa: division by a linear divisor, lane-scan forms (lindiv_kernels.h) against the scan_kernels.h form
OUT=gpurun_out/r03aa; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "scan_onepass_variants or horner_scan or kzg" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
B="timeout 150 python bench.py --no-cpu --workload open22 --steps 300 --warmup 30 --samples 5"
for m in 0 8d 8l 16d 16l; do
  RONK_LINDIV=$m $B > $OUT/open22_$m.json 2>> $OUT/err
done
export TMPDIR=/tmp
for m in 0 8d 16d 16l; do
  (cd /tmp && RONK_LINDIV=$m timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$m -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload open22 --steps 200 --warmup 20 --samples 2 > /dev/null 2>> $GRAFT_REPO_ROOT/$OUT/err)
  f=$(find /tmp/prof_$m -name "*kernel_stats.csv" | head -1)
  echo "== mode $m" >> $OUT/kernel_stats.txt; head -6 "$f" >> $OUT/kernel_stats.txt
done
tail -3 $OUT/err
cat $OUT/kernel_stats.txt | cut -c1-230
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
    print('%-20s %10.1f op/s (%.4f ms)  device %.2f us  frac %.3f verified %s' % ('$f'.split('/')[-1], d['value'], d['ms_per_step'], r.get('device_us_per_step') or 0, r['frac'], d.get('verified')))
except Exception as e: print('$f', 'ERR', e)
PY
done
