#!/bin/bash
# round 3, GPU session e: full parity suite on the feature / three-pass specialisations + their timings
OUT=gpurun_out/r03e; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
for cfg in 0 1; do
  E=""; [ $cfg = 0 ] && E="RONK_NO_CFG_KERNELS=1"
  env $E timeout 150 python bench.py --no-cpu --workload mul22 --steps 50 --samples 3 > $OUT/mul22_cfg$cfg.json 2>> $OUT/err
  env $E timeout 150 python bench.py --no-cpu --workload rs16 --steps 30 --warmup 5 --samples 3 > $OUT/rs16_cfg$cfg.json 2>> $OUT/err
  for lg in 23 24 26; do
    env $E timeout 150 python bench.py --no-cpu --mode streams --streams 1 --log2n $lg --steps 40 --warmup 5 --samples 3 --rotate 1 > $OUT/ntt${lg}_cfg$cfg.json 2>> $OUT/err
  done
done
timeout 200 python bench.py --workload e2e22 --steps 64 --samples 3 > $OUT/e2e22.json 2>> $OUT/err
timeout 100 python bench.py --no-cpu --workload sharded --ranks 8 --log2n 26 --steps 20 --warmup 3 > $OUT/sharded8.json 2>> $OUT/err
timeout 100 python bench.py --no-cpu --workload fourstep --log2n 26 --steps 20 --warmup 3 > $OUT/fourstep.json 2>> $OUT/err
tail -3 $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1])
    print('%-22s' % '$f'.split('/')[-1], {k:(round(d[k],4) if isinstance(d[k],float) else d[k]) for k in ('value','ms_per_step','verified') if k in d}, 'single', (d.get('single') or {}).get('ms_per_step'), 'lat_us', d['roofline'].get('device_us_per_step'))
except Exception as e: print('$f', 'ERR', e)
PY
done
