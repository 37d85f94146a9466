#!/bin/bash
# round 3, GPU session a: parity of the in-flight lanes + how the K transforms should reach the library (warm / cold)
OUT=gpurun_out/r03a; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 -k "in_flight or many_dev or batch16_of or full_vector or sixteen_rows or several_streams or planner_default or config3" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
B="timeout 150 python bench.py --no-cpu --steps 96 --warmup 16 --samples 3"
$B --mode streams --streams 2 > $OUT/streams2.json 2> $OUT/err
$B --mode streams --streams 1 > $OUT/streams1.json 2>> $OUT/err
$B --mode many --group 16 > $OUT/many16.json 2>> $OUT/err
$B --mode many --group 96 > $OUT/many96.json 2>> $OUT/err
$B --mode many --group 2 > $OUT/many2.json 2>> $OUT/err
$B --mode many --group 4 > $OUT/many4.json 2>> $OUT/err
$B --mode batch --group 2 > $OUT/batch2.json 2>> $OUT/err
$B --mode batch --group 4 > $OUT/batch4.json 2>> $OUT/err
$B --mode batch --group 16 > $OUT/batch16x22.json 2>> $OUT/err
RONK_IN_FLIGHT=1 $B --mode batch --group 16 > $OUT/batch16x22_nolanes.json 2>> $OUT/err
$B --mode many --group 16 --twf 22 > $OUT/many16_twf22.json 2>> $OUT/err
timeout 150 python bench.py --no-cpu --workload mul22 --steps 50 --samples 3 > $OUT/mul22.json 2>> $OUT/err
RONK_IN_FLIGHT=1 timeout 150 python bench.py --no-cpu --workload mul22 --steps 50 --samples 3 > $OUT/mul22_nolanes.json 2>> $OUT/err
timeout 150 python bench.py --steps 20 --warmup 5 > $OUT/driver_args.json 2>> $OUT/err
tail -5 $OUT/err
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']; w=d.get('warm') or {}
        print('%-28s value %9.1f ms/step %.4f | warm %9.1f | frac %.3f lat %.3f latwarm %s | lat_us %.2f verified %s' % (f.split('/')[-1], d['value'], d['ms_per_step'], w.get('value',0), r['frac'], r.get('frac_latency',0), r.get('frac_latency_warm'), r.get('device_us_per_step',0), d.get('verified')))
    except Exception as e: print(f, 'ERR', e)
PY
