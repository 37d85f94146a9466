#!/bin/bash
# round 3, GPU session i: tile width of the two-lane plans with the twiddle matrix; multiply on the FEAT KIND-3 kernels
OUT=gpurun_out/r03i; mkdir -p $OUT
B="timeout 150 python bench.py --no-cpu --steps 96 --warmup 16 --samples 5"
for lc in -1 1 2 3; do
  $B --mode many --tile-logc $lc > $OUT/many_lc$lc.json 2>> $OUT/err
done
$B --mode streams --streams 2 > $OUT/streams2.json 2>> $OUT/err
$B --mode streams --streams 3 > $OUT/streams3.json 2>> $OUT/err
timeout 150 python bench.py --no-cpu --workload mul22 --steps 50 --samples 5 > $OUT/mul22.json 2>> $OUT/err
RONK_TWF_MAX_LOG=18 timeout 150 python bench.py --no-cpu --workload mul22 --steps 50 --samples 5 > $OUT/mul22_twf18.json 2>> $OUT/err
timeout 150 python bench.py --no-cpu --workload mul22 --log2n 21 --steps 50 --samples 5 > $OUT/mul21.json 2>> $OUT/err
RONK_TWF_MAX_LOG=18 timeout 150 python bench.py --no-cpu --workload mul22 --log2n 21 --steps 50 --samples 5 > $OUT/mul21_twf18.json 2>> $OUT/err
timeout 150 python bench.py --steps 20 --warmup 5 > $OUT/driver_args.json 2>> $OUT/err
tail -3 $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; w=d.get('warm') or {}
    print('%-22s cold %9.1f (%.4f ms)  warm %9.1f  lat_us cold %.2f  frac %.3f frac_lat %.3f' % ('$f'.split('/')[-1], d['value'], d['ms_per_step'], w.get('value',0), r.get('device_us_per_step',0), r['frac'], r.get('frac_latency',0)))
except Exception as e: print('$f', 'ERR', e)
PY
done
