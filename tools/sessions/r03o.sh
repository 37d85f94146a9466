#!/bin/bash
# round 3, GPU session o: three-pass splits with the specialised kernels (pass 1 of 2^24 takes 126 us of 224: stride aliasing?)
OUT=gpurun_out/r03o; mkdir -p $OUT
B="timeout 150 python bench.py --no-cpu --mode streams --streams 1 --steps 40 --warmup 5 --samples 3"
for lg in 24 25 26; do
  $B --log2n $lg > $OUT/n${lg}_default.json 2>> $OUT/err
  for sp in "8,8" "9,8" "9,7" "8,9" "10,7" "10,8" "9,9" "10,6" "7,9" "11,7"; do
    ka=${sp%,*}; kb=${sp#*,}; kc=$((lg - ka - kb))
    if [ $kc -ge 4 ] && [ $kc -le 12 ]; then
      RONK_SPLIT3="$sp" $B --log2n $lg > $OUT/n${lg}_${ka}_${kb}_${kc}.json 2>> $OUT/err
    fi
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
