#!/bin/bash
# round 3, GPU session at: latency form vs tile kernels, more shapes around 2^19 / 2^20 coefficients in all
OUT=gpurun_out/r03at; mkdir -p $OUT
B="timeout 150 python bench.py --no-cpu --mode streams --streams 1 --steps 100 --warmup 10 --samples 3"
for sm in 0 1; do
  for cfg in "18 2" "17 4" "15 16" "14 32" "12 128" "17 8" "15 32" "14 64" "12 256" "10 1024" "18 1" "15 8" "13 32"; do
    set -- $cfg
    RONK_SMALL=$sm $B --log2n $1 --batch $2 > $OUT/b$2x$1_small$sm.json 2>> $OUT/err
  done
done
tail -2 $OUT/err
python - <<PY
import json,glob,re
rows={}
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        k=re.match(r'.*/(b\d+x\d+)_small(\d)\.json',f).groups()
        rows.setdefault(k[0],{})[k[1]]=r.get('device_us_per_step') or 0
    except Exception as e: print(f,'ERR',e)
for k,v in sorted(rows.items(), key=lambda kv: [int(x) for x in re.findall(r'\d+',kv[0])][::-1]):
    print('%-12s tile %.2f us  latency-form %.2f us  -> %s' % (k, v.get('0',0), v.get('1',0), 'latency' if v.get('1',9e9)<v.get('0',0) else 'tile'))
PY
