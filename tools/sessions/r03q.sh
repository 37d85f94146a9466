#!/bin/bash
# round 3, GPU session q: two passes instead of three at 2^23 / 2^24 (RONK_THREE_PASS_FROM=25), with and without the specialised bodies
OUT=gpurun_out/r03q; mkdir -p $OUT
B="timeout 150 python bench.py --no-cpu --mode streams --streams 1 --steps 40 --warmup 5 --samples 3"
for lg in 23 24; do
  $B --log2n $lg > $OUT/n${lg}_three.json 2>> $OUT/err
  RONK_THREE_PASS_FROM=25 $B --log2n $lg > $OUT/n${lg}_two.json 2>> $OUT/err
  RONK_THREE_PASS_FROM=25 RONK_TWF_MAX_LOG=24 $B --log2n $lg > $OUT/n${lg}_two_twf.json 2>> $OUT/err
  RONK_THREE_PASS_FROM=25 RONK_NO_CFG_KERNELS=1 $B --log2n $lg > $OUT/n${lg}_two_generic.json 2>> $OUT/err
done
RONK_SPLIT3="9,8" $B --log2n 24 > $OUT/n24_three_9_8_7.json 2>> $OUT/err
tail -3 $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
    print('%-24s %.4f ms  passes %s  verified %s' % ('$f'.split('/')[-1], d['ms_per_step'], [round(x,1) for x in (r.get('pass_us') or [])], d.get('verified')))
except Exception as e: print('$f', 'ERR', e)
PY
done
