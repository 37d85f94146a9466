#!/bin/bash
# round 3, GPU session g: four-step phases on the specialised bodies (KIND 4), sharded enqueue order; planner tile sweep (cold)
OUT=gpurun_out/r03g; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "sharded or dist or fourstep or four_step" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
for cfg in 0 1; do
  E=""; [ $cfg = 0 ] && E="RONK_NO_CFG_KERNELS=1"
  env $E timeout 100 python bench.py --no-cpu --workload sharded --ranks 8 --log2n 26 --steps 20 --warmup 3 > $OUT/sharded8_cfg$cfg.json 2>> $OUT/err
  env $E timeout 100 python bench.py --no-cpu --workload fourstep --log2n 26 --steps 20 --warmup 3 > $OUT/fourstep_cfg$cfg.json 2>> $OUT/err
done
# planner sweep, HBM-cold: one plan of batch B at 2^k, widest tile forced (RONK_MAX_LOGC) vs the planner's rule
B="timeout 150 python bench.py --no-cpu --warmup 8 --samples 3 --mode batch"
for spec in "22 16 64" "21 32 64" "20 64 128" "19 128 256"; do
  set -- $spec
  for lc in auto 4 3 2; do
    E=""; [ $lc != auto ] && E="RONK_MAX_LOGC=$lc"
    env $E $B --log2n $1 --group $2 --steps $3 > $OUT/sweep_$1_$lc.json 2>> $OUT/err
  done
done
tail -3 $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1])
    print('%-24s' % '$f'.split('/')[-1], 'value %10.1f  ms/step %.4f  warm %s' % (d['value'], d['ms_per_step'], (d.get('warm') or {}).get('value')))
except Exception as e: print('$f', 'ERR', e)
PY
done
