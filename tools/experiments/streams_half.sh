O=gpurun_out/r02ah; mkdir -p $O
run() { local name=$1; local envs=$2; shift 2
  env $envs timeout 120 python bench.py --no-cpu --no-verify "$@" > $O/$name.json 2>> $O/err || echo "FAIL $name" >> $O/err; }
run s2_c2_h0 RONK_HALF_LDS=0 --streams 2 --tile-logc 2 --steps 300 --warmup 30
for s in 2 3 4 6; do for c in 2 3; do
  run s${s}_c${c}_h1 RONK_HALF_LDS=1 --streams $s --tile-logc $c --steps 300 --warmup 30
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02ah/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'NTT/s %.0f'%d['value'], 'us/transform %.2f'%(d['ms_per_step']*1e3))
    except Exception as e: print(f, 'ERR', e)
PY
