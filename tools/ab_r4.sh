#!/bin/bash
# tools/ab_r4.sh <outdir> -- A/B of the R4 round structure (RONK_R4MID=0 / 1) on one box: one transform at a time for the sizes
# whose plans contain 2^9 / 2^10-row passes, the two-lane throughput regime at 2^20 / 2^21, batched shapes, the multiply at 2^21.
OUT=${1:-gpurun_out/ab_r4}; mkdir -p $OUT
for R4 in 0 1 0 1; do
  for lg in 18 19 20 21 24 25 26; do
    RONK_R4MID=$R4 timeout 200 python bench.py --no-cpu --mode streams --streams 1 --log2n $lg --steps 40 --warmup 5 --samples 3 2>>$OUT/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('r4=$R4 one-at-a-time 2^$lg  %.2f us cold  %s  verified %s' % (r['device_us_per_step'], ['%.1f'%p for p in (r.get('pass_us') or [])], d['verified']))" >> $OUT/ab.txt
  done
  for lg in 20 21; do
    RONK_R4MID=$R4 timeout 200 python bench.py --no-cpu --log2n $lg --steps 100 --warmup 10 --samples 3 2>>$OUT/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('r4=$R4 two lanes 2^$lg  %.1f NTT/s  %.2f us per transform' % (d['value'], d['ms_per_step']*1e3))" >> $OUT/ab.txt
  done
  for spec in "20 64" "18 256" "19 128"; do set -- $spec
    for H in 0 ""; do
    RONK_HALF_LDS=$H RONK_R4MID=$R4 timeout 200 python bench.py --no-cpu --workload batch16 --log2n $1 --batch $2 --steps 10 --warmup 2 --samples 3 2>>$OUT/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('r4=$R4 half=${H:-auto} batch $2 x 2^$1  %.4f ms  verified %s' % (d['ms_per_step'], d['verified']))" >> $OUT/ab.txt
    done
  done
done
sort -k2,9 -s $OUT/ab.txt > $OUT/ab_sorted.txt; cat $OUT/ab_sorted.txt
