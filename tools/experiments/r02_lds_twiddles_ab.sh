mkdir -p gpurun_out/r02p
timeout 120 ./gpurun_bin/ubench 3 > gpurun_out/r02p/ubench.txt 2>&1; grep -h "ABL=  0 \|ABL= 56\|ABL= 48\|ABL= 64" gpurun_out/r02p/ubench.txt
timeout 200 python -m pytest tests -m gpu -x -q --timeout 300 -k "tuned or config3 or big_batches" 2>&1 | tail -2
for i in 1 2; do timeout 100 python bench.py --no-cpu --streams 1 > gpurun_out/r02p/bench_s1_$i.json 2>> gpurun_out/r02p/err; done
timeout 100 python bench.py --no-cpu > gpurun_out/r02p/bench_s2.json 2>> gpurun_out/r02p/err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02p/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], 'value %.1f'%d['value'], 'dev_us %.2f'%r['device_us_per_step'], 'pass', r.get('pass_us'), d.get('verified'))
PY
