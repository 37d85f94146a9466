#!/bin/bash
# round 3, GPU session az: tile width of the two-lane plan under the driver's arguments (HBM-cold headline)
OUT=gpurun_out/r03az; mkdir -p $OUT
for i in 1 2 3; do
  timeout 200 python bench.py --no-cpu --gpus 1 --steps 20 --warmup 5 > $OUT/drv_default_$i.json 2>> $OUT/err
  timeout 200 python bench.py --no-cpu --gpus 1 --steps 20 --warmup 5 --tile-logc 3 > $OUT/drv_c8_$i.json 2>> $OUT/err
done
timeout 200 python bench.py --no-cpu > $OUT/def_default.json 2>> $OUT/err
timeout 200 python bench.py --no-cpu --tile-logc 3 > $OUT/def_c8.json 2>> $OUT/err
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); w=d.get('warm') or {}
    print('%-22s cold %9.1f  warm %9.1f' % (f.split('/')[-1], d['value'], w.get('value',0)))
PY
