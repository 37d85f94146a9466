#!/bin/bash
# round 3, GPU session as: latency form (ntt_small.h) vs tile kernels around the 2^19-coefficient threshold, after the load hoisting
OUT=gpurun_out/r03as; mkdir -p $OUT
B="timeout 150 python bench.py --no-cpu --mode streams --streams 1 --steps 100 --warmup 10 --samples 3"
for sm in 0 1; do
  for lg in 16 17 18 19 20; do RONK_SMALL=$sm $B --log2n $lg > $OUT/n${lg}_small$sm.json 2>> $OUT/err; done
  RONK_SMALL=$sm $B --log2n 16 --batch 8 > $OUT/b8x16_small$sm.json 2>> $OUT/err
  RONK_SMALL=$sm $B --log2n 16 --batch 16 > $OUT/b16x16_small$sm.json 2>> $OUT/err
  RONK_SMALL=$sm $B --log2n 13 --batch 64 > $OUT/b64x13_small$sm.json 2>> $OUT/err
  RONK_SMALL=$sm $B --log2n 13 --batch 128 > $OUT/b128x13_small$sm.json 2>> $OUT/err
done
tail -2 $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
    print('%-24s %10.1f %s (%.5f ms per step) device_us %.2f verified %s' % ('$f'.split('/')[-1], d['value'], d['unit'], d['ms_per_step'], r.get('device_us_per_step') or 0, d.get('verified')))
except Exception as e: print('$f', 'ERR', e)
PY
done
