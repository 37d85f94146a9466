#!/bin/bash
# round 3, GPU session al: evidence of the evaluate kernel (two Horner chains) -- rocprofv3 trace + PMC, bench line
OUT=gpurun_out/r03al; mkdir -p $OUT
bash tools/profile.sh eval22 r03_eval22 > $OUT/prof_eval22.txt 2>&1
cp gpurun_out/prof_r03_eval22/summary.txt $OUT/summary_eval22.txt; cp gpurun_out/prof_r03_eval22/summary.json $OUT/summary_eval22.json
cp $OUT/summary_eval22.json profiles/latest_pmc_eval22.json
timeout 150 python bench.py --no-cpu --workload eval22 > $OUT/bench_eval22.json 2>> $OUT/err
timeout 150 python bench.py --no-cpu --workload open22 > $OUT/bench_open22.json 2>> $OUT/err
timeout 300 python -m pytest tests -m gpu -q --timeout 600 -k "scan or horner or eval" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
head -5 $OUT/summary_eval22.txt | cut -c1-160
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], d['value'], d['ms_per_step'], r['frac'], r.get('traffic'), (r.get('valu') or {}).get('insts_per_coeff'), d.get('verified'))
PY
