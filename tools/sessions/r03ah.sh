#!/bin/bash
# round 3, GPU session ah: co-resident workgroups of one pass started a fraction of a round apart (in-kernel skew)
OUT=gpurun_out/r03ah; mkdir -p $OUT
B="timeout 150 python bench.py --no-cpu --mode streams --streams 1 --steps 96 --warmup 16 --samples 5"
for lc in 2 1; do
  for sk in 0 50 100 200 300 500; do
    RONK_WG_SKEW_TICKS=$sk $B --tile-logc $lc > $OUT/lat_c${lc}_skew$sk.json 2>> $OUT/err
  done
done
RONK_WG_SKEW_TICKS=0 $B > $OUT/lat_default.json 2>> $OUT/err
for sk in 0 100 300; do
  RONK_WG_SKEW_TICKS=$sk timeout 150 python bench.py --no-cpu --steps 96 --warmup 16 --samples 5 > $OUT/many_skew$sk.json 2>> $OUT/err
  RONK_WG_SKEW_TICKS=$sk timeout 150 python bench.py --no-cpu --mode batch --group 16 --steps 96 --warmup 16 --samples 5 > $OUT/b16_skew$sk.json 2>> $OUT/err
done
tail -2 $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; w=d.get('warm') or {}
    print('%-26s cold %10.1f (%.4f ms)  warm %9.1f  lat_us %.2f verified %s' % ('$f'.split('/')[-1], d['value'], d['ms_per_step'], w.get('value',0), r.get('device_us_per_step') or 0, d.get('verified')))
except Exception as e: print('$f', 'ERR', e)
PY
done
