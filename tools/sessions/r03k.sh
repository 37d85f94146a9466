#!/bin/bash
# round 3, GPU session k: half-image LDS exchanges (8 waves per SIMD) in the two-lane regime, with the twiddle matrix
OUT=gpurun_out/r03k; mkdir -p $OUT
B="timeout 150 python bench.py --no-cpu --steps 96 --warmup 16 --samples 5"
$B --mode many > $OUT/many_full.json 2>> $OUT/err
RONK_HALF_LDS=1 $B --mode many > $OUT/many_half_c4.json 2>> $OUT/err
RONK_HALF_LDS=1 $B --mode many --tile-logc 3 > $OUT/many_half_c8.json 2>> $OUT/err
RONK_HALF_LDS=1 $B --mode batch --group 16 > $OUT/batch16_half.json 2>> $OUT/err
RONK_HALF_LDS=1 RONK_MAX_LOGC=2 $B --mode batch --group 16 > $OUT/batch16_half_c4.json 2>> $OUT/err
tail -3 $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; w=d.get('warm') or {}
    print('%-22s cold %9.1f (%.4f ms)  warm %9.1f  lat_us cold %.2f  frac %.3f' % ('$f'.split('/')[-1], d['value'], d['ms_per_step'], w.get('value',0), r.get('device_us_per_step',0), r['frac']))
except Exception as e: print('$f', 'ERR', e)
PY
done
