#!/bin/bash
# round 3, GPU session ao: EARLY tile kernels (table twiddles and matrix column fetched ahead of the barriers) for passes with one workgroup per CU
OUT=gpurun_out/r03ao; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "all_sizes or planner or config3 or large_plans or determinism or device_pointer" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
B="timeout 150 python bench.py --no-cpu --mode streams --streams 1 --steps 96 --warmup 16 --samples 5"
for e in 0 1 0 1; do
  RONK_EARLY=$e $B > $OUT/lat22_early${e}_$RANDOM.json 2>> $OUT/err
done
for e in 0 1; do
  RONK_EARLY=$e $B --log2n 21 > $OUT/lat21_early$e.json 2>> $OUT/err
  RONK_EARLY=$e timeout 150 python bench.py --no-cpu --workload mul22 > $OUT/mul22_early$e.json 2>> $OUT/err
  RONK_EARLY=$e timeout 150 python bench.py --no-cpu --steps 96 --warmup 16 --samples 5 > $OUT/many_early$e.json 2>> $OUT/err
done
tail -2 $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; w=d.get('warm') or {}
    print('%-30s cold %10.1f (%.4f ms)  warm %9.1f  device_us %.2f  passes %s verified %s' % ('$f'.split('/')[-1], d['value'], d['ms_per_step'], w.get('value',0), r.get('device_us_per_step') or 0, [round(x,1) for x in (r.get('pass_us') or [])], d.get('verified')))
except Exception as e: print('$f', 'ERR', e)
PY
done
