#!/bin/bash
# round 3, GPU session au: evidence on the final sources (planner rule for the latency form changed): GPU suite, headline bench lines,
# rocprofv3 trace + PMC of the headline (both regimes), the batched shape and the multiply; the re-measured shapes with the rule in place
OUT=gpurun_out/r03au; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 500 bash tools/profile.sh ntt22 r03_1stream --mode streams --streams 1 > $OUT/prof_1stream.txt 2>&1
timeout 500 bash tools/profile.sh ntt22 r03_many > $OUT/prof_many.txt 2>&1
timeout 500 bash tools/profile.sh batch16 r03_batch16 > $OUT/prof_batch16.txt 2>&1
timeout 500 bash tools/profile.sh mul22 r03_mul22 > $OUT/prof_mul22.txt 2>&1
for t in 1stream many batch16 mul22; do cp gpurun_out/prof_r03_$t/summary.txt $OUT/summary_$t.txt; cp gpurun_out/prof_r03_$t/summary.json $OUT/summary_$t.json; done
cp $OUT/summary_1stream.json profiles/latest_pmc_ntt22.json; cp $OUT/summary_batch16.json profiles/latest_pmc_batch16.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/err
timeout 300 python bench.py > $OUT/bench_default.json 2>> $OUT/err
timeout 200 python bench.py --workload batch16 > $OUT/bench_batch16.json 2>> $OUT/err
B="timeout 150 python bench.py --no-cpu --mode streams --streams 1 --steps 100 --warmup 10 --samples 3"
for cfg in "19 1" "18 2" "17 4" "13 128" "14 64" "16 16"; do set -- $cfg; $B --log2n $1 --batch $2 > $OUT/rule_b$2x$1.json 2>> $OUT/err; done
tail -2 $OUT/err
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_*.json'))+sorted(glob.glob('$OUT/rule_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']; w=d.get('warm') or {}
        print('%-28s value %11.1f  ms/step %.5f  warm %9.1f  lat_us %8.2f  frac %.3f traffic %s valu %s verified %s' % (f.split('/')[-1], d['value'], d['ms_per_step'], w.get('value', 0), r.get('device_us_per_step') or 0, r['frac'], r.get('traffic'), (r.get('valu') or {}).get('insts_per_coeff'), d.get('verified')))
    except Exception as e: print(f,'ERR',e)
PY
