O=gpurun_out/r02ae; mkdir -p $O
run() { local name=$1; local envs=$2; shift 2
  env $envs timeout 120 python bench.py --no-cpu --no-verify "$@" > $O/$name.json 2>> $O/err || echo "FAIL $name" >> $O/err; }
for v in 0 1; do
  for lg in 18 19 20; do run f${lg}_s$v RONK_SMALL=$v --workload batch16 --log2n $lg --batch 1 --steps 400 --warmup 40; done
  run f16x8_s$v RONK_SMALL=$v --workload batch16 --log2n 16 --batch 8 --steps 400 --warmup 40
  run f14x32_s$v RONK_SMALL=$v --workload batch16 --log2n 14 --batch 32 --steps 400 --warmup 40
  run f14x64_s$v RONK_SMALL=$v --workload batch16 --log2n 14 --batch 64 --steps 400 --warmup 40
  run f13x64_s$v RONK_SMALL=$v --workload batch16 --log2n 13 --batch 64 --steps 400 --warmup 40
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02ae/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], 'us/step %.2f'%(d['ms_per_step']*1e3), 'dev_us %.2f'%r.get('device_us_per_step',0))
    except Exception as e: print(f, 'ERR', e)
PY
