O=gpurun_out/r02y; mkdir -p $O
for lg in 16 18 20; do timeout 300 python bench.py --workload msm20 --log2n $lg --no-cpu --steps 5 --samples 3 > $O/msm$lg.json 2>> $O/err; done
for c in 12 13 14 16; do RONK_MSM_C=$c timeout 300 python bench.py --workload msm20 --log2n 20 --no-cpu --no-verify --steps 5 --samples 3 > $O/msm20_c$c.json 2>> $O/err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02y/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'Mpoints/s %.2f'%(d['value']/1e6), 'ms/step %.3f'%d['ms_per_step'], d.get('verified'))
    except Exception as e: print(f, 'ERR', e)
PY
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_msm -o msm -- python $GRAFT_REPO_ROOT/bench.py --workload msm20 --no-cpu --no-verify --steps 3 --samples 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/prof_msm -name "*kernel_stats.csv" | head -1); head -12 "$f" > $O/rocprof_msm20_stats.csv; cat $O/rocprof_msm20_stats.csv | cut -c1-160
tail -3 $O/err
