#!/bin/bash
# round 3, GPU session t: three-pass splits with a SHORT first pass (its strided reads run at ~2.5 TB/s whatever the tile)
OUT=gpurun_out/r03t; mkdir -p $OUT
B="timeout 150 python bench.py --no-cpu --mode streams --streams 1 --steps 40 --warmup 5 --samples 3"
for sp in "6,9" "7,8" "6,8" "7,9" "5,9" "6,10"; do
  ka=${sp%,*}; kb=${sp#*,}
  RONK_SPLIT3="$sp" $B --log2n 24 > $OUT/n24_${ka}_${kb}.json 2>> $OUT/err
done
for sp in "7,9" "7,10" "6,10" "8,10" "6,9"; do
  ka=${sp%,*}; kb=${sp#*,}
  RONK_SPLIT3="$sp" $B --log2n 26 > $OUT/n26_${ka}_${kb}.json 2>> $OUT/err
done
for f in $OUT/*.json; do python - <<PY
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
print('%-18s %.4f ms  passes %s  verified %s' % ('$f'.split('/')[-1], d['ms_per_step'], [round(x,1) for x in (r.get('pass_us') or [])], d.get('verified')))
PY
done
