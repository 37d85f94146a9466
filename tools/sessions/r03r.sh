#!/bin/bash
OUT=gpurun_out/r03r; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "tuned_tile or all_sizes or planner_default or config3_variant or sharded" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
B="timeout 150 python bench.py --no-cpu --mode streams --streams 1 --steps 40 --warmup 5 --samples 3"
for lg in 23 24; do $B --log2n $lg > $OUT/n$lg.json 2>> $OUT/err; done
for f in $OUT/*.json; do python - <<PY
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
print('%-12s %.4f ms  passes %s  verified %s' % ('$f'.split('/')[-1], d['ms_per_step'], [round(x,1) for x in (r.get('pass_us') or [])], d.get('verified')))
PY
done
