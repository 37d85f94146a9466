#!/bin/bash
# round 3, GPU session af: two lanes started half a pass apart (RONK_LANE_SKEW_US) -- throughput of the headline
OUT=gpurun_out/r03af; mkdir -p $OUT
B="timeout 150 python bench.py --no-cpu --steps 96 --warmup 16 --samples 5"
for sk in 0 6 10 14 18 24 30 40; do
  RONK_LANE_SKEW_US=$sk $B > $OUT/many_skew$sk.json 2>> $OUT/err
done
for sk in 0 14 24; do
  RONK_LANE_SKEW_US=$sk $B --tile-logc 3 > $OUT/many_c8_skew$sk.json 2>> $OUT/err
done
tail -2 $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; w=d.get('warm') or {}
    print('%-26s cold %10.1f (%.4f ms)  warm %9.1f  verified %s' % ('$f'.split('/')[-1], d['value'], d['ms_per_step'], w.get('value',0), d.get('verified')))
except Exception as e: print('$f', 'ERR', e)
PY
done
