#!/bin/bash
# round 3, GPU session aj: scans -- workgroup sum through the cross-lane network (one barrier), Y^b from three tables
OUT=gpurun_out/r03aj; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "scan or horner or kzg or div or open or eval or lagrange" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
for wl in eval22 open22; do
  timeout 150 python bench.py --no-cpu --workload $wl > $OUT/bench_${wl}_nopmc.json 2>> $OUT/err
done
RONK_LINDIV=0 timeout 150 python bench.py --no-cpu --workload open22 > $OUT/bench_open22_old_form.json 2>> $OUT/err
RONK_NO_ONEPASS_SCANS=1 timeout 150 python bench.py --no-cpu --workload eval22 > $OUT/bench_eval22_two_launches.json 2>> $OUT/err
bash tools/profile.sh eval22 r03_eval22 > $OUT/prof_eval22.txt 2>&1
cp gpurun_out/prof_r03_eval22/summary.txt $OUT/summary_eval22.txt; cp gpurun_out/prof_r03_eval22/summary.json $OUT/summary_eval22.json
cp $OUT/summary_eval22.json profiles/latest_pmc_eval22.json
timeout 150 python bench.py --no-cpu --workload eval22 > $OUT/bench_eval22.json 2>> $OUT/err
tail -2 $OUT/err; head -8 $OUT/summary_eval22.txt | cut -c1-200
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], d['value'], d['ms_per_step'], r['frac'], r.get('traffic'), (r.get('valu') or {}).get('insts_per_coeff'), d.get('verified'))
PY
