#!/bin/bash
# round 3, GPU session w: config 4 (1024 x 2^16) with wider tiles (generic bodies for a like-for-like comparison)
OUT=gpurun_out/r03w; mkdir -p $OUT
B="timeout 150 python bench.py --no-cpu --workload batch16 --steps 30 --warmup 5 --samples 5"
$B > $OUT/cfg4_default.json 2>> $OUT/err
for lc in 4 5 6; do
  RONK_MAX_LOGC=$lc RONK_NO_CFG_KERNELS=1 $B > $OUT/cfg4_generic_lc$lc.json 2>> $OUT/err
done
RONK_NO_CFG_KERNELS=1 $B > $OUT/cfg4_generic_default.json 2>> $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
print('%-28s %.4f ms  passes %s  verified %s' % ('$f'.split('/')[-1], d['ms_per_step'], [round(x,1) for x in (r.get('pass_us') or [])], d.get('verified')))
PY
done
