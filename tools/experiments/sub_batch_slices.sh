O=gpurun_out/r02ab; mkdir -p $O
run() { local name=$1; local envs=$2; shift 2
  env $envs timeout 120 python bench.py --no-cpu "$@" > $O/$name.json 2>> $O/err || echo "FAIL $name" >> $O/err; }
for m in 0 16 32 64 128 192; do
  run b16_m$m RONK_SUB_BATCH_MIB=$m --workload batch16 --steps 50 --warmup 10
done
for m in 0 32 64 128; do
  run b20_m$m RONK_SUB_BATCH_MIB=$m --workload batch16 --log2n 20 --batch 64 --steps 50 --warmup 10 --no-verify
  run b18_m$m RONK_SUB_BATCH_MIB=$m --workload batch16 --log2n 18 --batch 256 --steps 50 --warmup 10 --no-verify
  run b22_m$m RONK_SUB_BATCH_MIB=$m --workload batch16 --log2n 22 --batch 16 --steps 30 --warmup 5 --no-verify
  run b14_m$m RONK_SUB_BATCH_MIB=$m --workload batch16 --log2n 14 --batch 4096 --steps 50 --warmup 10 --no-verify
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02ab/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], 'value %.1f'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'frac %.3f'%r['frac'], d.get('verified'))
    except Exception as e: print(f, 'ERR', e)
PY
tail -3 $O/err
