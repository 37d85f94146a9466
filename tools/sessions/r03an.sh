#!/bin/bash
# round 3, GPU session an: division kernels with every load of a launch issued up front; evidence refresh for open22
OUT=gpurun_out/r03an; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "scan or horner or kzg or div or open" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
for i in 1 2; do timeout 150 python bench.py --no-cpu --workload open22 > $OUT/bench_open22_nopmc_$i.json 2>> $OUT/err; done
bash tools/profile.sh open22 r03_open22 > $OUT/prof_open22.txt 2>&1
cp gpurun_out/prof_r03_open22/summary.txt $OUT/summary_open22.txt; cp gpurun_out/prof_r03_open22/summary.json $OUT/summary_open22.json
cp $OUT/summary_open22.json profiles/latest_pmc_open22.json
timeout 150 python bench.py --no-cpu --workload open22 > $OUT/bench_open22.json 2>> $OUT/err
head -6 $OUT/summary_open22.txt | cut -c1-160
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], d['value'], d['ms_per_step'], r['frac'], r.get('traffic'), (r.get('valu') or {}).get('insts_per_coeff'), d.get('verified'))
PY
