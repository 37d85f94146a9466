#!/bin/bash
# round 3, GPU session am: latency kernel (ntt_small.h) with every global load of a pass issued up front
OUT=gpurun_out/r03am; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "all_sizes or config2 or device_pointer or batched or mul or conv or rs_ or lde or small or tuned or planner or dft" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout 200 python bench.py --no-cpu --workload roundtrip16 > $OUT/bench_roundtrip16.json 2>> $OUT/err
for lg in 13 16 17 18 19; do
  timeout 150 python bench.py --no-cpu --mode streams --streams 1 --log2n $lg --steps 200 --warmup 20 --samples 5 > $OUT/bench_ntt$lg.json 2>> $OUT/err
done
timeout 150 python bench.py --no-cpu --workload rs16 > $OUT/bench_rs16.json 2>> $OUT/err
tail -2 $OUT/err
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print('%-26s %12.1f %s  ms/step %.5f  device_us %.2f  verified %s' % (f.split('/')[-1], d['value'], d['unit'], d['ms_per_step'], r.get('device_us_per_step') or 0, d.get('verified')))
    except Exception as e: print(f,'ERR',e)
PY
