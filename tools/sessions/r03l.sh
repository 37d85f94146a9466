#!/bin/bash
# round 3, GPU session l: which passes of big batches should use the half-image exchanges (RONK_HALF_LDS: unset = rule, 0 never, 1 always, 2 row passes only)
OUT=gpurun_out/r03l; mkdir -p $OUT
B="timeout 150 python bench.py --no-cpu --warmup 8 --samples 5 --mode batch"
for spec in "22 16 64" "21 32 64" "20 64 128" "19 128 256"; do
  set -- $spec
  for h in rule 0 1 2; do
    E=""; [ $h != rule ] && E="RONK_HALF_LDS=$h"
    env $E $B --log2n $1 --group $2 --steps $3 > $OUT/b_$1_half$h.json 2>> $OUT/err
  done
done
for h in rule 0 1 2; do
  E=""; [ $h != rule ] && E="RONK_HALF_LDS=$h"
  env $E timeout 150 python bench.py --no-cpu --workload batch16 --steps 30 --warmup 5 --samples 5 > $OUT/cfg4_half$h.json 2>> $OUT/err
  env $E timeout 150 python bench.py --no-cpu --workload mul22 --steps 50 --samples 5 > $OUT/mul22_half$h.json 2>> $OUT/err
done
tail -3 $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; w=d.get('warm') or {}
    print('%-22s cold %11.1f (%.4f ms)  warm %9.1f' % ('$f'.split('/')[-1], d['value'], d['ms_per_step'], w.get('value',0)))
except Exception as e: print('$f', 'ERR', e)
PY
done
