O=gpurun_out/r02w; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q --timeout 600 -k "scan or horner or divrem or linear" 2>&1 | tail -5
run() { local name=$1; local envs=$2; shift 2
  env $envs timeout 120 python bench.py --no-cpu "$@" > $O/$name.json 2>> $O/err || echo "FAIL $name" >> $O/err; }
for v in new old; do
  E=RONK_DUMMY=1; [ $v = old ] && E=RONK_NO_ONEPASS_SCANS=1
  run open22_$v $E --workload open22 --steps 200 --warmup 20
  run eval22_$v $E --workload eval22 --steps 200 --warmup 20
  run open20_$v $E --workload open22 --log2n 20 --steps 200 --warmup 20
  run eval20_$v $E --workload eval22 --log2n 20 --steps 200 --warmup 20
  run open23_$v $E --workload open22 --log2n 23 --steps 100 --warmup 20
  run eval24_$v $E --workload eval22 --log2n 24 --steps 100 --warmup 20
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02w/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], 'value %.1f'%d['value'], 'ms/step %.5f'%d['ms_per_step'], 'frac %.3f'%r['frac'], d.get('verified'))
    except Exception as e: print(f, 'ERR', e)
PY
tail -5 $O/err
