O=gpurun_out/r02u; mkdir -p $O
run() { local name=$1; local envs=$2; shift 2
  env $envs timeout 120 python bench.py --no-cpu --no-verify "$@" > $O/$name.json 2>> $O/err || echo "FAIL $name" >> $O/err; }
for h in 0 1 2; do
  run b16_h$h RONK_HALF_LDS=$h --workload batch16 --steps 50 --warmup 10
  run rs16_h$h RONK_HALF_LDS=$h --workload rs16 --steps 50 --warmup 10
  run b17_h$h RONK_HALF_LDS=$h --workload batch16 --log2n 17 --batch 512 --steps 50 --warmup 10
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02u/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], 'value %.1f'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'frac %.3f'%r['frac'], 'pass', r.get('pass_us'))
    except Exception as e: print(f, 'ERR', e)
PY
tail -3 $O/err
