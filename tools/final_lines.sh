#!/bin/bash
# tools/final_lines.sh <outdir> -- every bench workload once on the library as built (no rocprof, no counters): the numbers of the
# round's evidence run re-read on the FINAL library in a few minutes.  One compact table on stdout, the JSON lines under <outdir>.
OUT=${1:-gpurun_out/final_lines}
mkdir -p $OUT
run() { name=$1; shift; timeout 300 python bench.py "$@" 2>> $OUT/err | tail -1 > $OUT/$name.json; }
run ntt22_driver_args --gpus 1 --steps 20 --warmup 5
run ntt22_default
run ntt22_1stream --no-cpu --mode streams --streams 1
run mont_driver_args --no-cpu --prime 0xFFFFFFFC00000001 --steps 20 --warmup 5
run mont_mul22 --no-cpu --workload mul22 --prime 0xFFFFFFFC00000001
for wl in mul22 batch16 roundtrip16 rs16 open22 eval22 vecmul24; do run $wl --no-cpu --workload $wl; done
run msm20 --no-cpu --workload msm20 --log2n 20 --steps 5 --samples 3
for lg in 20 23 24 26; do run ntt$lg --no-cpu --mode streams --streams 1 --log2n $lg --steps 40 --warmup 5 --samples 3; done
run fourstep_1gpu --no-cpu --workload fourstep --log2n 26 --steps 20 --warmup 3
run sharded_8ranks_1gpu --no-cpu --workload sharded --ranks 8 --log2n 26 --steps 20 --warmup 3
RONK_BENCH_BACKEND=gloo run two_ranks_gloo_smoke --gpus 2 --steps 20 --warmup 5 --no-cpu
python - $OUT <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        d = json.load(open(f))
    except Exception as e:   # noqa: BLE001
        print("%-24s NO LINE (%s)" % (os.path.basename(f)[:-5], e)); continue
    r = d.get("roofline") or {}
    print("%-24s %14.4f %-10s ms_per_step %.5f  frac %s  verified %s" % (os.path.basename(f)[:-5], d["value"], d["unit"][:10], d["ms_per_step"],
          ("%.3f" % r["frac"]) if r.get("frac") is not None else "-", d.get("verified")))
PY
