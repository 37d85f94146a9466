#!/bin/bash
# round 3, GPU session m: single-pass plans (n <= 2^12, 2^26 coefficients per call) on the whole-polynomial bodies (KIND 5) vs generic
OUT=gpurun_out/r03m; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 -k "all_sizes or batched_ragged or single or small or kat or golden or poly" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
for lg in 12 11 10 9 8 7 6; do
  b=$((1 << (26 - lg)))
  for cfg in 0 1; do
    E=""; [ $cfg = 0 ] && E="RONK_NO_CFG_KERNELS=1"
    env $E timeout 150 python bench.py --no-cpu --workload batch16 --log2n $lg --batch $b --steps 30 --warmup 5 --samples 3 > $OUT/sp${lg}_cfg$cfg.json 2>> $OUT/err
  done
done
tail -3 $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
    print('%-18s %.4f ms per 2^26 coefficients  frac %.3f  verified %s' % ('$f'.split('/')[-1], d['ms_per_step'], r['frac'], d.get('verified')))
except Exception as e: print('$f', 'ERR', e)
PY
done
