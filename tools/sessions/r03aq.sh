#!/bin/bash
# round 3, GPU session aq: two lanes with the half-image kernels (four workgroups of 512 work-items per CU instead of two)
OUT=gpurun_out/r03aq; mkdir -p $OUT
B="timeout 150 python bench.py --no-cpu --steps 96 --warmup 16 --samples 5"
for h in -1 1 2; do
  if [ $h = -1 ]; then $B > $OUT/many_halfdefault.json 2>> $OUT/err; else RONK_HALF_LDS=$h $B > $OUT/many_half$h.json 2>> $OUT/err; fi
done
RONK_HALF_LDS=1 $B --tile-logc 3 > $OUT/many_c8_half1.json 2>> $OUT/err
RONK_HALF_LDS=1 $B --tile-logc 1 > $OUT/many_c2_half1.json 2>> $OUT/err
$B --tile-logc 1 > $OUT/many_c2_halfdefault.json 2>> $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; w=d.get('warm') or {}
    print('%-30s cold %10.1f (%.4f ms)  warm %9.1f verified %s' % ('$f'.split('/')[-1], d['value'], d['ms_per_step'], w.get('value',0), d.get('verified')))
except Exception as e: print('$f', 'ERR', e)
PY
done
