#!/bin/bash
# round 3, GPU session ar: two-level inter-pass twiddles issued before the previous group's stores (scheduling fence), full-image kernels
OUT=gpurun_out/r03ar; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -k "all_sizes or large_plans or planner or sharded or dist or fourstep" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
B="timeout 150 python bench.py --no-cpu --mode streams --streams 1 --steps 40 --warmup 5 --samples 3"
for h in d 0; do
  if [ $h = d ]; then E=""; else E="RONK_HALF_LDS=0"; fi
  for lg in 19 20 23 24 26; do env $E $B --log2n $lg > $OUT/n${lg}_half$h.json 2>> $OUT/err; done
  env $E timeout 100 python bench.py --no-cpu --workload fourstep --log2n 26 --steps 20 --warmup 3 > $OUT/fourstep_half$h.json 2>> $OUT/err
done
tail -2 $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
    print('%-24s %10.1f %s (%.4f ms)  passes %s verified %s' % ('$f'.split('/')[-1], d['value'], d['unit'], d['ms_per_step'], [round(x,1) for x in (r.get('pass_us') or [])], d.get('verified')))
except Exception as e: print('$f', 'ERR', e)
PY
done
