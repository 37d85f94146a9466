#!/bin/bash
# round 3, GPU session ap: full GPU suite and the headline bench lines on the final library
OUT=gpurun_out/r03ap; mkdir -p $OUT
timeout 1400 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/err
timeout 300 python bench.py > $OUT/bench_default.json 2>> $OUT/err
timeout 300 python bench.py --workload roundtrip16 > $OUT/bench_roundtrip16.json 2>> $OUT/err
timeout 300 python bench.py --workload rs16 > $OUT/bench_rs16.json 2>> $OUT/err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']; w=d.get('warm') or {}
    print('%-28s value %11.1f  ms/step %.4f  warm %9.1f  lat_us %8.2f  frac %.3f  verified %s  cpu %s' % (f.split('/')[-1], d['value'], d['ms_per_step'], w.get('value', 0), r.get('device_us_per_step') or 0, r['frac'], d.get('verified'), (d.get('cpu_baseline') or {}).get('value')))
PY
