#!/bin/bash
# tools/profile_all.sh <tag> -- the round's evidence run on the GPU box (through gpurun): parity suite, bench lines of every
# workload (driver arguments for the headline), rocprofv3 trace + PMC of the headline (one transform at a time, and the
# two-lane throughput regime), the multiply and the batched shape.
TAG=${1:-r04}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
# counters first: the bench lines below then carry the per-step traffic / VALU sums of THIS library (bench.py refuses counter
# files taken on other sources)
timeout 500 bash tools/profile.sh ntt22 ${TAG}_1stream --mode streams --streams 1 > $OUT/prof_1stream.txt 2>&1
timeout 500 bash tools/profile.sh ntt22 ${TAG}_many > $OUT/prof_many.txt 2>&1
timeout 500 bash tools/profile.sh batch16 ${TAG}_batch16 > $OUT/prof_batch16.txt 2>&1
timeout 500 bash tools/profile.sh mul22 ${TAG}_mul22 > $OUT/prof_mul22.txt 2>&1
for t in 1stream many batch16 mul22; do cp gpurun_out/prof_${TAG}_$t/summary.txt $OUT/summary_$t.txt; cp gpurun_out/prof_${TAG}_$t/summary.json $OUT/summary_$t.json; done
cp gpurun_out/prof_${TAG}_1stream/summary.json profiles/latest_pmc_ntt22.json; cp gpurun_out/prof_${TAG}_batch16/summary.json profiles/latest_pmc_batch16.json
for wl in open22 eval22; do timeout 300 bash tools/profile.sh $wl ${TAG}_$wl > $OUT/prof_$wl.txt 2>&1; cp gpurun_out/prof_${TAG}_$wl/summary.json profiles/latest_pmc_$wl.json; cp gpurun_out/prof_${TAG}_$wl/summary.json $OUT/summary_$wl.json; cp gpurun_out/prof_${TAG}_$wl/summary.txt $OUT/summary_$wl.txt; done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/err
timeout 300 python bench.py > $OUT/bench_default.json 2>> $OUT/err
timeout 200 python bench.py --no-cpu --mode streams --streams 2 > $OUT/bench_ntt22_2streams.json 2>> $OUT/err
timeout 200 python bench.py --no-cpu --mode batch --group 16 > $OUT/bench_ntt22_batch16.json 2>> $OUT/err
for wl in batch16 mul22 roundtrip16 rs16; do timeout 300 python bench.py --workload $wl > $OUT/bench_$wl.json 2>> $OUT/err; done
for wl in open22 eval22 vecmul24 vecadd24; do timeout 100 python bench.py --no-cpu --workload $wl > $OUT/bench_$wl.json 2>> $OUT/err; done
timeout 300 python bench.py --workload e2e22 --steps 64 --samples 3 > $OUT/bench_e2e22.json 2>> $OUT/err
for lg in 16 20; do timeout 300 python bench.py --workload msm20 --log2n $lg --steps 5 --samples 3 > $OUT/bench_msm$lg.json 2>> $OUT/err; done
for lg in 20 21 23 24 26; do timeout 150 python bench.py --no-cpu --mode streams --streams 1 --log2n $lg --steps 40 --warmup 5 --samples 3 > $OUT/bench_ntt$lg.json 2>> $OUT/err; done
timeout 100 python bench.py --no-cpu --workload fourstep --log2n 26 --steps 20 --warmup 3 > $OUT/bench_fourstep_1gpu.json 2>> $OUT/err
timeout 100 python bench.py --no-cpu --workload sharded --ranks 8 --log2n 26 --steps 20 --warmup 3 > $OUT/bench_sharded_8ranks_1gpu.json 2>> $OUT/err
# the multi-rank control flow of bench.py (two ranks sharing this GPU over gloo: a smoke test of --gpus N, not a measurement)
# (plain `python bench.py --gpus 2`: bench.py starts its ranks itself)
RONK_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu > $OUT/bench_2ranks_gloo_smoke.json 2>> $OUT/err
RONK_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --workload fourstep --log2n 24 --steps 10 --warmup 2 --no-cpu > $OUT/bench_fourstep_2ranks_gloo_smoke.json 2>> $OUT/err
RONK_MUL_FUSED=0 timeout 300 python bench.py --no-cpu --workload mul22 > $OUT/bench_mul22_four_launches.json 2>> $OUT/err
tail -2 $OUT/err
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']; w=d.get('warm') or {}
        print('%-34s value %11.1f  ms/step %.4f  warm %9.1f  lat_us %8.2f  frac %.3f  lat %s  verified %s  cpu %s' % (f.split('/')[-1], d['value'], d['ms_per_step'], w.get('value', 0), r.get('device_us_per_step') or 0, r['frac'], r.get('frac_latency'), d.get('verified'), (d.get('cpu_baseline') or {}).get('value')))
    except Exception as e: print(f, 'ERR', e)
PY
