#!/bin/bash
# round 3, GPU session b: persistent software-pipelined tile kernels (RONK_PIPE) -- parity and A/B timing
OUT=gpurun_out/r03b; mkdir -p $OUT
RONK_PIPE=1 timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 -k "planner_default or in_flight or batch16_of or config4 or batched_ragged or many_dev or rs_encode or lde" > $OUT/pytest_pipe1.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_pipe1.log; tail -4 $OUT/pytest_pipe1.log
B="timeout 150 python bench.py --no-cpu --steps 96 --warmup 16 --samples 3"
for pipe in 0 1; do
  RONK_PIPE=$pipe RONK_IN_FLIGHT=1 $B --mode batch --group 16 > $OUT/b16_pipe${pipe}_lanes1.json 2>> $OUT/err
  RONK_PIPE=$pipe RONK_IN_FLIGHT=2 $B --mode batch --group 16 > $OUT/b16_pipe${pipe}_lanes2.json 2>> $OUT/err
  RONK_PIPE=$pipe RONK_IN_FLIGHT=1 $B --mode batch --group 4 > $OUT/b4_pipe${pipe}_lanes1.json 2>> $OUT/err
  RONK_PIPE=$pipe RONK_IN_FLIGHT=1 RONK_MAX_LOGC=3 $B --mode batch --group 16 > $OUT/b16_pipe${pipe}_c8.json 2>> $OUT/err
  RONK_PIPE=$pipe timeout 150 python bench.py --no-cpu --workload batch16 --steps 30 --warmup 5 --samples 3 > $OUT/cfg4_pipe${pipe}.json 2>> $OUT/err
  RONK_PIPE=$pipe RONK_IN_FLIGHT=1 timeout 150 python bench.py --no-cpu --workload mul22 --steps 50 --samples 3 > $OUT/mul22_pipe${pipe}.json 2>> $OUT/err
  RONK_PIPE=$pipe timeout 150 python bench.py --no-cpu --workload rs16 --steps 30 --warmup 5 --samples 3 > $OUT/rs16_pipe${pipe}.json 2>> $OUT/err
done
tail -5 $OUT/err
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']; w=d.get('warm') or {}
        print('%-28s value %9.1f ms/step %.4f | warm %9.1f | frac %.3f lat %.3f | lat_us %.2f verified %s' % (f.split('/')[-1], d['value'], d['ms_per_step'], w.get('value',0), r['frac'], r.get('frac_latency',0), r.get('device_us_per_step',0), d.get('verified')))
    except Exception as e: print(f, 'ERR', e)
PY
