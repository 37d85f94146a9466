#!/bin/bash
# round 3, GPU session p: wider tiles (32 / 64 columns) for the strided first pass of three-pass plans (generic bodies: shape test only)
OUT=gpurun_out/r03p; mkdir -p $OUT
B="timeout 150 python bench.py --no-cpu --mode streams --streams 1 --steps 40 --warmup 5 --samples 3"
for lg in 24 26; do
  for lc in 4 5 6; do
    RONK_MAX_LOGC=$lc RONK_NO_CFG_KERNELS=1 $B --log2n $lg > $OUT/n${lg}_lc$lc.json 2>> $OUT/err
  done
done
tail -3 $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
    print('%-22s %.4f ms  passes %s  verified %s' % ('$f'.split('/')[-1], d['ms_per_step'], [round(x,1) for x in (r.get('pass_us') or [])], d.get('verified')))
except Exception as e: print('$f', 'ERR', e)
PY
done
