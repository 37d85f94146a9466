#!/bin/bash
# tools/profile_refresh.sh <round tag> -- after a kernel-source change late in a round: re-take the rocprofv3 trace + PMC files
# bench.py reads (profiles/latest_pmc_*.json carry the hash of the kernel sources) and the headline bench lines; the rest of the
# round's evidence (tools/profile_all.sh) stays as it is.  Output: gpurun_out/refresh_<tag>/.
TAG=${1:-r05}
OUT=gpurun_out/refresh_$TAG
mkdir -p $OUT
prof() {
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
cp profiles/latest_pmc_*.json $OUT/
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/err
timeout 300 python bench.py > $OUT/bench_default.json 2>> $OUT/err
timeout 200 python bench.py --no-cpu --mode streams --streams 1 > $OUT/bench_ntt22_1stream.json 2>> $OUT/err
timeout 300 python bench.py --prime 0xFFFFFFFC00000001 --steps 20 --warmup 5 > $OUT/bench_mont_driver_args.json 2>> $OUT/err
for wl in batch16 mul22; do timeout 300 python bench.py --no-cpu --workload $wl > $OUT/bench_$wl.json 2>> $OUT/err; done
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print('%-34s value %11.1f  ms/step %.4f  frac %.3f  traffic %s  valu %s  verified %s' % (f.split('/')[-1], d['value'], d['ms_per_step'], r['frac'], r.get('traffic'), (r.get('valu') or {}).get('insts_per_coeff'), d.get('verified')))
    except Exception as e: print(f, 'ERR', e)
PY
