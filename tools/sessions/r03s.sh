#!/bin/bash
# round 3, GPU session s: the strided first pass of the three-pass plans -- occupancy (half images) and tile width
OUT=gpurun_out/r03s; mkdir -p $OUT
B="timeout 150 python bench.py --no-cpu --mode streams --streams 1 --steps 40 --warmup 5 --samples 3"
for lg in 24 26; do
  for h in rule 0 1; do
    E=""; [ $h != rule ] && E="RONK_HALF_LDS=$h"
    env $E $B --log2n $lg > $OUT/n${lg}_half$h.json 2>> $OUT/err
  done
  for lc in 3 2; do RONK_MAX_LOGC=$lc $B --log2n $lg > $OUT/n${lg}_lc$lc.json 2>> $OUT/err; done
done
for f in $OUT/*.json; do python - <<PY
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
print('%-18s %.4f ms  passes %s  verified %s' % ('$f'.split('/')[-1], d['ms_per_step'], [round(x,1) for x in (r.get('pass_us') or [])], d.get('verified')))
PY
done
