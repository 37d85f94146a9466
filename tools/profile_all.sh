#!/bin/bash
# tools/profile_all.sh <tag> -- the round's evidence run on the GPU box (through gpurun): parity suite, bench lines of every
# workload (driver arguments for the headline), rocprofv3 trace + PMC of the headline in both configurations.
TAG=${1:-r02}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/err
timeout 200 python bench.py > $OUT/bench_default.json 2>> $OUT/err
timeout 100 python bench.py --no-cpu --streams 1 > $OUT/bench_ntt22_1stream.json 2>> $OUT/err
for wl in batch16 mul22 roundtrip16 rs16; do timeout 300 python bench.py --workload $wl > $OUT/bench_$wl.json 2>> $OUT/err; done
for wl in open22 eval22 vecmul24 vecadd24; do timeout 100 python bench.py --no-cpu --workload $wl > $OUT/bench_$wl.json 2>> $OUT/err; done
for lg in 16 18 20; do timeout 300 python bench.py --workload msm20 --log2n $lg --steps 5 --samples 3 > $OUT/bench_msm$lg.json 2>> $OUT/err; done
timeout 100 python bench.py --no-cpu --workload fourstep --log2n 26 --steps 20 --warmup 3 > $OUT/bench_fourstep_1gpu.json 2>> $OUT/err
timeout 100 python bench.py --no-cpu --workload sharded --ranks 8 --log2n 26 --steps 20 --warmup 3 > $OUT/bench_sharded_8ranks_1gpu.json 2>> $OUT/err
timeout 400 bash tools/profile.sh ntt22 ${TAG}_1stream --streams 1 > $OUT/prof_1stream.txt 2>&1
timeout 400 bash tools/profile.sh ntt22 ${TAG}_2streams --streams 2 > $OUT/prof_2streams.txt 2>&1
timeout 400 bash tools/profile.sh batch16 ${TAG}_batch16 > $OUT/prof_batch16.txt 2>&1
timeout 400 bash tools/profile.sh open22 ${TAG}_open22 > $OUT/prof_open22.txt 2>&1
timeout 400 bash tools/profile.sh eval22 ${TAG}_eval22 > $OUT/prof_eval22.txt 2>&1
tail -2 $OUT/err
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f.split('/')[-1], 'value %.1f'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'dev_us %.2f'%r.get('device_us_per_step',0), 'frac %.3f'%r['frac'], 'verified', d.get('verified'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e)
PY
