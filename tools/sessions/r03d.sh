#!/bin/bash
# round 3, GPU session d: host pipeline, per-peer copy streams / RCCL exchange, Rust FFI replay, e2e bench
OUT=gpurun_out/r03d; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x --timeout 400 -k "sharded or rust_ffi or host_mirror or planner_default or config4 or in_flight or many_dev or several_streams or dist" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -6 $OUT/pytest.log
timeout 200 python bench.py --workload e2e22 --steps 64 --samples 3 > $OUT/e2e22.json 2> $OUT/err
RONK_HOST_NO_PIPELINE=1 timeout 200 python bench.py --workload e2e22 --steps 64 --samples 3 > $OUT/e2e22_nopipe.json 2>> $OUT/err
timeout 200 python bench.py --workload e2e22 --steps 64 --samples 3 --group 32 > $OUT/e2e22_g32.json 2>> $OUT/err
timeout 100 python bench.py --no-cpu --workload sharded --ranks 8 --log2n 26 --steps 20 --warmup 3 > $OUT/sharded8.json 2>> $OUT/err
timeout 100 python bench.py --no-cpu --workload sharded --ranks 1 --log2n 26 --steps 20 --warmup 3 > $OUT/sharded1.json 2>> $OUT/err
tail -3 $OUT/err
for f in $OUT/*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','verified') if k in d}, d.get('single'), d['roofline'].get('frac'))
"; done
