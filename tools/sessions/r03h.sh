#!/bin/bash
# round 3, GPU session h: full inter-pass twiddle matrix at 2^22 (RONK_TWF_MAX_LOG=22) against the two-level tables, A/B/A/B
OUT=gpurun_out/r03h; mkdir -p $OUT
B="timeout 150 python bench.py --no-cpu --steps 96 --warmup 16 --samples 5"
for rep in 1 2; do
  for twf in 18 22; do
    RONK_TWF_MAX_LOG=$twf $B --mode many > $OUT/many_twf${twf}_$rep.json 2>> $OUT/err
    RONK_TWF_MAX_LOG=$twf $B --mode batch --group 16 > $OUT/batch16_twf${twf}_$rep.json 2>> $OUT/err
    RONK_TWF_MAX_LOG=$twf timeout 150 python bench.py --no-cpu --workload mul22 --steps 50 --samples 5 > $OUT/mul22_twf${twf}_$rep.json 2>> $OUT/err
  done
done
for twf in 18 20 21; do
  RONK_TWF_MAX_LOG=$twf $B --mode streams --streams 1 --log2n 20 --rotate 8 > $OUT/n20_twf$twf.json 2>> $OUT/err
  RONK_TWF_MAX_LOG=$twf $B --mode streams --streams 1 --log2n 21 --rotate 8 > $OUT/n21_twf$twf.json 2>> $OUT/err
done
tail -3 $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; w=d.get('warm') or {}
    print('%-26s cold %9.1f (%.4f ms)  warm %9.1f  lat_us cold %.2f warm %s' % ('$f'.split('/')[-1], d['value'], d['ms_per_step'], w.get('value',0), r.get('device_us_per_step',0), ('%.2f' % (16.0*(1<<d['config'].get('log2n',22))/ (r['frac_latency_warm']*8e12)*1e6)) if r.get('frac_latency_warm') else '-'))
except Exception as e: print('$f', 'ERR', e)
PY
done
