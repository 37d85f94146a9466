#!/bin/bash
# round 3, GPU session ae: lane-scan division as the default -- parity (every scan form), bench line, rocprofv3 trace + PMC
OUT=gpurun_out/r03ae; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "scan or horner or kzg or div or open or host_mirror or ffi_replay" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout 150 python bench.py --no-cpu --workload open22 > $OUT/bench_open22_nopmc.json 2>> $OUT/err
bash tools/profile.sh open22 r03_open22 > $OUT/prof_open22.txt 2>&1
cp gpurun_out/prof_r03_open22/summary.txt $OUT/summary_open22.txt; cp gpurun_out/prof_r03_open22/summary.json $OUT/summary_open22.json
cp $OUT/summary_open22.json profiles/latest_pmc_open22.json
timeout 150 python bench.py --no-cpu --workload open22 > $OUT/bench_open22.json 2>> $OUT/err
timeout 150 python bench.py --no-cpu --workload eval22 > $OUT/bench_eval22.json 2>> $OUT/err
tail -2 $OUT/err; head -30 $OUT/summary_open22.txt | cut -c1-200
python - <<PY
import json
for f in ('bench_open22_nopmc','bench_open22','bench_eval22'):
    d=json.loads(open('$OUT/%s.json'%f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], r['frac'], r.get('traffic'), (r.get('valu') or {}).get('insts_per_coeff'), d.get('verified'))
PY
