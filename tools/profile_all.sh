#!/bin/bash
# tools/profile_all.sh <round tag> -- the round's evidence run on the GPU box (through gpurun): parity suite, rocprofv3 trace +
# PMC of the headline in BOTH regimes (the two-lane throughput regime `value` is quoted on, and one transform at a time), of the
# Montgomery-prime transform, the multiply, the batched shape and the scans, then the bench lines of every workload (driver
# arguments for the headline).  Everything lands under gpurun_out/final_<tag>/ (merged back by gpurun); the summaries are
# named <tag>_<what>_rocprof.{txt,json} -- the name bench.py quotes as `source` -- and are committed under profiles/ by that name.
TAG=${1:-r05}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
# counters first: the bench lines below then carry the per-step traffic / VALU sums of THIS library (bench.py refuses counter
# files taken on other sources).  latest_pmc_ntt22.json = the kernels `value` runs (two lanes, 4-column tiles).
prof() {   # prof <workload> <what> <latest name or -> [bench args]
  local wl=$1 what=$2 latest=$3; shift 3
  timeout 500 bash tools/profile.sh $wl ${TAG}_$what "$@" > $OUT/prof_$what.log 2>&1
  cp gpurun_out/prof_${TAG}_$what/summary.txt $OUT/${TAG}_${what}_rocprof.txt
  cp gpurun_out/prof_${TAG}_$what/summary.json $OUT/${TAG}_${what}_rocprof.json
  [ "$latest" != "-" ] && cp gpurun_out/prof_${TAG}_$what/summary.json profiles/$latest
}
prof ntt22 ntt22_many latest_pmc_ntt22.json
prof ntt22 ntt22_1stream latest_pmc_ntt22_1stream.json --mode streams --streams 1
prof ntt22 ntt22_mont latest_pmc_ntt22_mont.json --prime 0xFFFFFFFC00000001
prof batch16 batch16 latest_pmc_batch16.json
prof mul22 mul22 -
prof open22 open22 latest_pmc_open22.json
prof eval22 eval22 latest_pmc_eval22.json
cp profiles/latest_pmc_*.json $OUT/
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/err
timeout 300 python bench.py > $OUT/bench_default.json 2>> $OUT/err
timeout 200 python bench.py --no-cpu --mode streams --streams 1 > $OUT/bench_ntt22_1stream.json 2>> $OUT/err
timeout 200 python bench.py --no-cpu --mode streams --streams 2 > $OUT/bench_ntt22_2streams.json 2>> $OUT/err
timeout 200 python bench.py --no-cpu --mode batch --group 16 > $OUT/bench_ntt22_batch16.json 2>> $OUT/err
timeout 300 python bench.py --prime 0xFFFFFFFC00000001 --steps 20 --warmup 5 > $OUT/bench_mont_driver_args.json 2>> $OUT/err
timeout 200 python bench.py --no-cpu --prime 0xFFFFFFFC00000001 --mode streams --streams 1 > $OUT/bench_mont_1stream.json 2>> $OUT/err
timeout 200 python bench.py --no-cpu --prime 0x3a00000000000001 > $OUT/bench_mont_29x2p57.json 2>> $OUT/err
for wl in mul22 batch16 roundtrip16; do timeout 300 python bench.py --no-cpu --workload $wl --prime 0xFFFFFFFC00000001 > $OUT/bench_mont_$wl.json 2>> $OUT/err; done
for wl in batch16 mul22 roundtrip16 rs16; do timeout 300 python bench.py --workload $wl > $OUT/bench_$wl.json 2>> $OUT/err; done
for wl in open22 eval22 vecmul24 vecadd24; do timeout 100 python bench.py --no-cpu --workload $wl > $OUT/bench_$wl.json 2>> $OUT/err; done
timeout 300 python bench.py --workload e2e22 --steps 64 --samples 3 > $OUT/bench_e2e22.json 2>> $OUT/err
for lg in 16 20; do timeout 300 python bench.py --workload msm20 --log2n $lg --steps 5 --samples 3 > $OUT/bench_msm$lg.json 2>> $OUT/err; done
for lg in 20 21 23 24 26; do timeout 150 python bench.py --no-cpu --mode streams --streams 1 --log2n $lg --steps 40 --warmup 5 --samples 3 > $OUT/bench_ntt$lg.json 2>> $OUT/err; done
timeout 100 python bench.py --no-cpu --workload fourstep --log2n 26 --steps 20 --warmup 3 > $OUT/bench_fourstep_1gpu.json 2>> $OUT/err
timeout 100 python bench.py --no-cpu --workload sharded --ranks 8 --log2n 26 --steps 20 --warmup 3 > $OUT/bench_sharded_8ranks_1gpu.json 2>> $OUT/err
# the multi-rank control flow of bench.py (two ranks sharing this GPU over gloo: a smoke test of --gpus N, not a measurement)
RONK_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu > $OUT/bench_2ranks_gloo_smoke.json 2>> $OUT/err
RONK_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --workload fourstep --log2n 24 --steps 10 --warmup 2 --no-cpu > $OUT/bench_fourstep_2ranks_gloo_smoke.json 2>> $OUT/err
timeout 300 python tools/mul_sizes.py > $OUT/mul_sizes.txt 2>> $OUT/err
FUZZ_SEED=51 FUZZ_SECONDS=150 timeout 400 python tools/fuzz_gpu.py mont ntt mul > $OUT/fuzz_mont.txt 2>&1
FUZZ_SEED=61 FUZZ_SECONDS=200 timeout 500 python tools/fuzz_gpu.py lindiv divrem vec sharded > $OUT/fuzz_callers.txt 2>&1
tail -2 $OUT/err
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']; w=d.get('warm') or {}
        print('%-34s value %11.1f  ms/step %.4f  warm %9.1f  lat_us %8.2f  frac %.3f  lat %s  traffic %s  verified %s  cpu %s' % (f.split('/')[-1], d['value'], d['ms_per_step'], w.get('value', 0), r.get('device_us_per_step') or 0, r['frac'], r.get('frac_latency'), r.get('traffic'), d.get('verified'), (d.get('cpu_baseline') or {}).get('value')))
    except Exception as e: print(f, 'ERR', e)
PY
