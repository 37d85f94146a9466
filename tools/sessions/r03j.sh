#!/bin/bash
OUT=gpurun_out/r03j; mkdir -p $OUT
for f in 18 22; do for i in 18 22; do
  RONK_MUL_FWD_TWF=$f RONK_MUL_INV_TWF=$i timeout 150 python bench.py --no-cpu --workload mul22 --steps 50 --samples 5 > $OUT/mul22_f${f}_i$i.json 2>> $OUT/err
  RONK_MUL_FWD_TWF=$f RONK_MUL_INV_TWF=$i timeout 150 python bench.py --no-cpu --workload mul22 --log2n 21 --steps 50 --samples 5 > $OUT/mul21_f${f}_i$i.json 2>> $OUT/err
done; done
tail -3 $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('%-22s %.4f ms  verified %s' % ('$f'.split('/')[-1], d['ms_per_step'], d.get('verified')))
PY
done
