O=gpurun_out/r02x; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | tail -3
for w in open22 eval22; do timeout 120 python bench.py --no-cpu --workload $w --steps 200 --warmup 20 > $O/$w.json 2>> $O/err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02x/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], 'value %.1f'%d['value'], 'dev_us %.2f'%r['device_us_per_step'], 'frac %.3f'%r['frac'], d.get('verified'))
PY
