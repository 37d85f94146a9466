#!/bin/bash
# round 3, GPU session n: software prefetch of upcoming inputs into the Infinity Cache in the two-lane regime (RONK_PREFETCH = distance)
OUT=gpurun_out/r03n; mkdir -p $OUT
B="timeout 150 python bench.py --no-cpu --steps 96 --warmup 16 --samples 5 --mode many"
$B > $OUT/pf0.json 2>> $OUT/err
for d in 1 2 3 4; do RONK_PREFETCH=$d $B > $OUT/pf$d.json 2>> $OUT/err; done
for g in 64 128 512 1024; do RONK_PREFETCH=2 RONK_PREFETCH_GRID=$g $B > $OUT/pf2_g$g.json 2>> $OUT/err; done
$B > $OUT/pf0_again.json 2>> $OUT/err
tail -3 $OUT/err
for f in $OUT/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; w=d.get('warm') or {}
    print('%-18s cold %9.1f (%.4f ms)  warm %9.1f  verified %s' % ('$f'.split('/')[-1], d['value'], d['ms_per_step'], w.get('value',0), d.get('verified')))
except Exception as e: print('$f', 'ERR', e)
PY
done
