#!/bin/bash
# round 3, GPU session ak: scans -- two Horner chains (half the dependent depth), wavefront carries one product deep
OUT=gpurun_out/r03ak; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "scan or horner or kzg or div or open or eval" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
for i in 1 2; do for wl in eval22 open22; do
  timeout 150 python bench.py --no-cpu --workload $wl > $OUT/bench_${wl}_$i.json 2>> $OUT/err
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], d['value'], d['ms_per_step'], r['frac'], d.get('verified'))
PY
