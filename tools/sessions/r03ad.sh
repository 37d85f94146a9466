#!/bin/bash
# round 3, GPU session ad: lane-scan division -- resident workgroups per CU (rounds per launch)
OUT=gpurun_out/r03ad; mkdir -p $OUT
B="timeout 150 python bench.py --no-cpu --workload open22 --steps 300 --warmup 30 --samples 5"
for cfg in "8d 0 0" "8d 4 4" "8d 5 5" "8d 6 6" "8d 3 3" "8d 8 4" "8d 4 8" "8d 2 2" "8l 4 4" "16d 2 2" "16d 3 3" "16l 2 2"; do
  set -- $cfg
  RONK_LINDIV=$1 RONK_LINDIV_OCC1=$2 RONK_LINDIV_OCC2=$3 $B > $OUT/open22_$1_$2_$3.json 2>> $OUT/err
done
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
    print('%-24s %10.1f op/s (%.4f ms)  device %.2f us  frac %.3f verified %s' % ('$f'.split('/')[-1], d['value'], d['ms_per_step'], r.get('device_us_per_step') or 0, r['frac'], d.get('verified')))
except Exception as e: print('$f', 'ERR', e)
PY
done
