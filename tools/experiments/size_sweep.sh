mkdir -p gpurun_out/r02n
for k in 4 5 6 7 8 9 10 11 12; do b=$((1 << (26 - k))); timeout 100 python bench.py --no-cpu --workload batch16 --log2n $k --batch $b --steps 100 --warmup 10 > gpurun_out/r02n/bench_single_pass_2p$k.json 2>> gpurun_out/r02n/err; done
for k in 18 20; do b=$((1 << (26 - k))); timeout 100 python bench.py --no-cpu --workload batch16 --log2n $k --batch $b --steps 100 --warmup 10 > gpurun_out/r02n/bench_batched_2p$k.json 2>> gpurun_out/r02n/err; done
for k in 24 25 26; do timeout 100 python bench.py --no-cpu --no-verify --workload batch16 --log2n $k --batch 1 --steps 50 --warmup 5 > gpurun_out/r02n/bench_single_2p$k.json 2>> gpurun_out/r02n/err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02n/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f.split('/')[-1], 'value %.1f'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'frac %.3f'%r['frac'], d.get('verified'))
    except Exception as e: print(f, 'ERR', e)
PY
