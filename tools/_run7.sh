mkdir -p gpurun_out/r02i
timeout 200 python -m pytest tests -m gpu -x -q --timeout 300 -k "scan or horner or linear_divisor or callers" > gpurun_out/r02i/pytest_scan.log 2>&1; tail -3 gpurun_out/r02i/pytest_scan.log
for wl in open22 eval22; do
timeout 100 python bench.py --no-cpu --workload $wl > gpurun_out/r02i/bench_${wl}_new.json 2>> gpurun_out/r02i/err
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02i/prof_open -- python $GRAFT_REPO_ROOT/bench.py --workload open22 --steps 100 --warmup 10 --samples 1 --no-cpu --no-verify > /dev/null 2>&1
RONK_NO_FUSED_SCANS=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02i/prof_open_old -- python $GRAFT_REPO_ROOT/bench.py --workload open22 --steps 100 --warmup 10 --samples 1 --no-cpu --no-verify > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,glob,sqlite3
for f in sorted(glob.glob('gpurun_out/r02i/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f, 'value %.1f'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'frac %.3f'%r['frac'], d.get('verified'))
    except Exception as e: print(f, 'ERR', e)
for d in ('prof_open','prof_open_old'):
    fs=glob.glob('gpurun_out/r02i/%s/**/*_results.db'%d, recursive=True)
    if fs:
        db=sqlite3.connect(fs[0])
        for row in db.execute("select name,total_calls,average from top_kernels"): print(d, row)
PY
find gpurun_out/r02i -name "*.db" -delete
