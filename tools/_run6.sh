mkdir -p gpurun_out/r02h
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 > gpurun_out/r02h/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02h/pytest_gpu.log; tail -25 gpurun_out/r02h/pytest_gpu.log
for wl in open22 eval22; do
RONK_NO_FUSED_SCANS=1 timeout 100 python bench.py --no-cpu --workload $wl > gpurun_out/r02h/bench_${wl}_old.json 2>> gpurun_out/r02h/err
timeout 100 python bench.py --no-cpu --workload $wl > gpurun_out/r02h/bench_${wl}_new.json 2>> gpurun_out/r02h/err
done
timeout 100 python bench.py --no-cpu --workload sharded --ranks 8 --log2n 26 --steps 20 --warmup 3 > gpurun_out/r02h/bench_sharded_8ranks_1gpu.json 2>> gpurun_out/r02h/err
timeout 100 python bench.py --no-cpu --workload sharded --ranks 8 --log2n 26 --steps 20 --warmup 3 --chunks 1 > gpurun_out/r02h/bench_sharded_8ranks_1gpu_c1.json 2>> gpurun_out/r02h/err
tail -3 gpurun_out/r02h/err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02h/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f, 'value %.1f'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'frac %.3f'%r['frac'], d.get('verified'))
    except Exception as e: print(f, 'ERR', e)
PY
