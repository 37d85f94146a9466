# experiment A (round 2, session 3): TileCfg::HALF (two-phase 32-bit LDS exchanges) A/B on one box
O=gpurun_out/r02s; mkdir -p $O
run() { # name, env, args...
  local name=$1; local envs=$2; shift 2
  env $envs timeout 120 python bench.py --no-cpu --no-verify "$@" > $O/$name.json 2>> $O/err || echo "FAIL $name" >> $O/err
}
for h in 0 1; do
  run b16_h$h RONK_HALF_LDS=$h --workload batch16 --steps 50 --warmup 10
  run b16_lc3_h$h RONK_HALF_LDS=$h --workload batch16 --steps 50 --warmup 10 --tile-logc 3
  run n22_s1_h$h RONK_HALF_LDS=$h --streams 1 --steps 200 --warmup 20
  run n22_s2_c2_h$h RONK_HALF_LDS=$h --streams 2 --tile-logc 2 --steps 200 --warmup 20
  run n22_s2_c3_h$h RONK_HALF_LDS=$h --streams 2 --tile-logc 3 --steps 200 --warmup 20
  run n22_s4_c2_h$h RONK_HALF_LDS=$h --streams 4 --tile-logc 2 --steps 200 --warmup 20
  run b20_h$h RONK_HALF_LDS=$h --workload batch16 --log2n 20 --batch 64 --steps 50 --warmup 10
  run b18_h$h RONK_HALF_LDS=$h --workload batch16 --log2n 18 --batch 256 --steps 50 --warmup 10
  run b22_h$h RONK_HALF_LDS=$h --workload batch16 --log2n 22 --batch 16 --steps 30 --warmup 5
  run mul22_h$h RONK_HALF_LDS=$h --workload mul22 --steps 100 --warmup 10
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02s/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], 'value %.1f'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'frac %.3f'%r['frac'], 'pass', r.get('pass_us'), d.get('verified'))
    except Exception as e: print(f, 'ERR', e)
PY
RONK_HALF_LDS=1 timeout 300 python -m pytest tests -m gpu -x -q --timeout 300 -k "not dist and not sharded" 2>&1 | tail -3
tail -5 $O/err
