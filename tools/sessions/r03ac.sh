#!/bin/bash
# round 3, GPU session ac: lane-scan division -- direct loads with LDS stores (d), all direct (D), all LDS (l); sums fetched eight at a time
OUT=gpurun_out/r03ac; mkdir -p $OUT
for m in 8d 16d; do RONK_LINDIV=$m timeout 300 python tests/scan_subprocess_check.py > $OUT/check_$m.log 2>&1; echo "$m check rc $?"; done
B="timeout 150 python bench.py --no-cpu --workload open22 --steps 300 --warmup 30 --samples 5"
for m in 0 8l 8d 8D 16l 16d; do
  RONK_LINDIV=$m $B > $OUT/open22_$m.json 2>> $OUT/err
done
export TMPDIR=/tmp
for m in 8l 8d 16l 16d; do
  (cd /tmp && RONK_LINDIV=$m timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace_$m/trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-verify --workload open22 --steps 100 --warmup 10 --samples 1 > /dev/null 2>> $GRAFT_REPO_ROOT/$OUT/err)
  python tools/rocprof_summary.py $OUT/trace_$m $OUT/trace_$m/summary x.txt > /dev/null 2>&1
  echo "== $m"; grep -E "lindiv|chunk_sum" $OUT/trace_$m/summary.txt | head -3 | cut -c1-170
  find $OUT/trace_$m -name "*.db" -delete 2>/dev/null
done
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
    print('%-20s %10.1f op/s (%.4f ms)  device %.2f us  frac %.3f verified %s' % ('$f'.split('/')[-1], d['value'], d['ms_per_step'], r.get('device_us_per_step') or 0, r['frac'], d.get('verified')))
except Exception as e: print('$f', 'ERR', e)
PY
done
