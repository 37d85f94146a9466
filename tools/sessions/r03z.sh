#!/bin/bash
# round 3, GPU session z: persistent prefetching kernels, second attempt (round twiddles staged in LDS once per workgroup)
OUT=gpurun_out/r03z; mkdir -p $OUT
RONK_PIPE=1 timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 -k "planner_default or batch16_of or config4 or batched_ragged or all_sizes" > $OUT/pytest_pipe1.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_pipe1.log; tail -3 $OUT/pytest_pipe1.log
B="timeout 150 python bench.py --no-cpu --steps 96 --warmup 16 --samples 5"
for pipe in 0 1; do
  RONK_PIPE=$pipe $B --mode batch --group 16 > $OUT/b16_pipe$pipe.json 2>> $OUT/err
  RONK_PIPE=$pipe RONK_HALF_LDS=0 $B --mode batch --group 16 > $OUT/b16_nohalf_pipe$pipe.json 2>> $OUT/err
  RONK_PIPE=$pipe timeout 150 python bench.py --no-cpu --workload batch16 --steps 30 --warmup 5 --samples 5 > $OUT/cfg4_pipe$pipe.json 2>> $OUT/err
  RONK_PIPE=$pipe $B --mode streams --streams 1 --log2n 24 --steps 40 > $OUT/n24_pipe$pipe.json 2>> $OUT/err
  RONK_PIPE=$pipe $B --mode many --tile-logc 3 > $OUT/many_c8_pipe$pipe.json 2>> $OUT/err
done
tail -2 $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; w=d.get('warm') or {}
    print('%-26s cold %10.1f (%.4f ms)  warm %9.1f  passes %s  verified %s' % ('$f'.split('/')[-1], d['value'], d['ms_per_step'], w.get('value',0), [round(x,1) for x in (r.get('pass_us') or [])], d.get('verified')))
except Exception as e: print('$f', 'ERR', e)
PY
done
