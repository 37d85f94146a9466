#!/bin/bash
# round 3, GPU session u: scatter the tile order of the strided first pass (RONK_TILE_SCATTER="pass:multiplier")
OUT=gpurun_out/r03u; mkdir -p $OUT
B="timeout 150 python bench.py --no-cpu --mode streams --streams 1 --steps 40 --warmup 5 --samples 3"
for lg in 24 26 22; do
  $B --log2n $lg > $OUT/n${lg}_nat.json 2>> $OUT/err
  for mul in 3 17 129 1025 40503 2654435761; do
    RONK_TILE_SCATTER="0:$mul" $B --log2n $lg > $OUT/n${lg}_p0_m$mul.json 2>> $OUT/err
  done
done
RONK_TILE_SCATTER="1:40503" $B --log2n 24 > $OUT/n24_p1_m40503.json 2>> $OUT/err
RONK_TILE_SCATTER="2:40503" $B --log2n 24 > $OUT/n24_p2_m40503.json 2>> $OUT/err
tail -2 $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
print('%-26s %.4f ms  passes %s  verified %s' % ('$f'.split('/')[-1], d['ms_per_step'], [round(x,1) for x in (r.get('pass_us') or [])], d.get('verified')))
PY
done
