O=gpurun_out/r02v; mkdir -p $O
run() { local name=$1; local envs=$2; shift 2
  env $envs timeout 120 python bench.py --no-cpu --no-verify "$@" > $O/$name.json 2>> $O/err || echo "FAIL $name" >> $O/err; }
for h in 0 1 d; do
  E=RONK_HALF_LDS=$h; [ $h = d ] && E=RONK_DUMMY=1
  run b16_h$h $E --workload batch16 --steps 50 --warmup 10
  run b19_h$h $E --workload batch16 --log2n 19 --batch 128 --steps 50 --warmup 10
  run b18_h$h $E --workload batch16 --log2n 18 --batch 256 --steps 50 --warmup 10
  run b20_h$h $E --workload batch16 --log2n 20 --batch 64 --steps 50 --warmup 10
  run b21_h$h $E --workload batch16 --log2n 21 --batch 32 --steps 50 --warmup 10
  run b22_h$h $E --workload batch16 --log2n 22 --batch 16 --steps 30 --warmup 5
  run b14_h$h $E --workload batch16 --log2n 14 --batch 4096 --steps 50 --warmup 10
  run b13_h$h $E --workload batch16 --log2n 13 --batch 8192 --steps 50 --warmup 10
done
run n22_s1_hd RONK_DUMMY=1 --streams 1 --steps 200 --warmup 20
run n22_s2_hd RONK_DUMMY=1 --streams 2 --steps 200 --warmup 20
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02v/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], 'value %.1f'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'frac %.3f'%r['frac'], 'pass', r.get('pass_us'))
    except Exception as e: print(f, 'ERR', e)
PY
timeout 300 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -2
tail -3 $O/err
