#!/bin/bash
# round 3, GPU session av: bench lines of the final library with the current counter files in place (traffic / VALU sums over the library's kernels only)
OUT=gpurun_out/r03av; mkdir -p $OUT
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/err
timeout 300 python bench.py > $OUT/bench_default.json 2>> $OUT/err
timeout 200 python bench.py --workload batch16 > $OUT/bench_batch16.json 2>> $OUT/err
timeout 200 python bench.py --workload mul22 > $OUT/bench_mul22.json 2>> $OUT/err
timeout 100 python bench.py --no-cpu --workload open22 > $OUT/bench_open22.json 2>> $OUT/err
timeout 100 python bench.py --no-cpu --workload eval22 > $OUT/bench_eval22.json 2>> $OUT/err
timeout 200 python bench.py --no-cpu --mode streams --streams 1 > $OUT/bench_ntt22_1stream.json 2>> $OUT/err
tail -2 $OUT/err
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']; w=d.get('warm') or {}
        print('%-28s value %11.1f  ms/step %.5f  warm %9.1f  lat_us %8.2f  frac %.3f traffic %s valu %s verified %s' % (f.split('/')[-1], d['value'], d['ms_per_step'], w.get('value', 0), r.get('device_us_per_step') or 0, r['frac'], r.get('traffic'), (r.get('valu') or {}).get('insts_per_coeff'), d.get('verified')))
    except Exception as e: print(f,'ERR',e)
PY
