#!/bin/bash
# round 3, GPU session y: inter-pass twiddles fetched one group ahead (no load behind a store on the in-order vmcnt)
OUT=gpurun_out/r03y; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "all_sizes or tuned_tile or planner_default or config3 or in_flight or sharded or dist or batched" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
B="timeout 150 python bench.py --no-cpu --steps 96 --warmup 16 --samples 5"
$B --mode many > $OUT/many.json 2>> $OUT/err
$B --mode batch --group 16 > $OUT/batch16x22.json 2>> $OUT/err
for lg in 19 20 21 22 23 24 26; do $B --mode streams --streams 1 --log2n $lg --steps 40 > $OUT/n$lg.json 2>> $OUT/err; done
timeout 150 python bench.py --no-cpu --workload mul22 --steps 50 --samples 5 > $OUT/mul22.json 2>> $OUT/err
timeout 150 python bench.py --no-cpu --workload batch16 --steps 30 --warmup 5 --samples 5 > $OUT/cfg4.json 2>> $OUT/err
timeout 100 python bench.py --no-cpu --workload fourstep --log2n 26 --steps 20 --warmup 3 > $OUT/fourstep.json 2>> $OUT/err
tail -2 $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; w=d.get('warm') or {}
    print('%-18s cold %10.1f (%.4f ms)  warm %9.1f  lat_us %.2f  passes %s' % ('$f'.split('/')[-1], d['value'], d['ms_per_step'], w.get('value',0), r.get('device_us_per_step') or 0, [round(x,1) for x in (r.get('pass_us') or [])]))
except Exception as e: print('$f', 'ERR', e)
PY
done
