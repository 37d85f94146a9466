O=gpurun_out/r02ad; mkdir -p $O
run() { local name=$1; local envs=$2; shift 2
  env $envs timeout 120 python bench.py --no-cpu "$@" > $O/$name.json 2>> $O/err || echo "FAIL $name" >> $O/err; }
for v in 0 1; do
  run rt16_s$v RONK_SMALL=$v --workload roundtrip16 --steps 400 --warmup 40
  for lg in 13 14 15 17; do run rt${lg}_s$v RONK_SMALL=$v --workload roundtrip16 --log2n $lg --steps 400 --warmup 40; done
  run rt18_s$v RONK_SMALL=$v --workload roundtrip16 --log2n 18 --steps 400 --warmup 40
  run b16x4_s$v RONK_SMALL=$v --workload batch16 --log2n 16 --batch 4 --steps 400 --warmup 40
  run b16x16_s$v RONK_SMALL=$v --workload batch16 --log2n 16 --batch 16 --steps 400 --warmup 40
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02ad/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], 'value %.1f'%d['value'], 'us/step %.2f'%(d['ms_per_step']*1e3), 'dev_us %.2f'%r.get('device_us_per_step',0), d.get('verified'))
    except Exception as e: print(f, 'ERR', e)
PY
timeout 600 python -m pytest tests -m gpu -x -q --timeout 600 -k "ntt or fft or roundtrip or config or oracle_all or inplace or determinism" 2>&1 | tail -3
tail -3 $O/err
