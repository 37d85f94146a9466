O=gpurun_out/r02ag; mkdir -p $O
for lg in 14 16 18 20 22; do for d in 9 8 7 6 5 4; do c=$((lg-d)); [ $c -lt 6 ] && continue; [ $c -gt 16 ] && continue
  RONK_MSM_C=$c timeout 300 python bench.py --workload msm20 --log2n $lg --no-cpu --no-verify --steps 5 --samples 3 > $O/msm${lg}_c$c.json 2>> $O/err; done; done
python - <<'PY'
import json,glob,re
rows={}
for f in sorted(glob.glob('gpurun_out/r02ag/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); m=re.search(r'msm(\d+)_c(\d+)',f)
        rows.setdefault(int(m.group(1)),[]).append((int(m.group(2)), d['ms_per_step']))
    except Exception as e: print(f,'ERR',e)
for lg in sorted(rows): print(lg, ' '.join('c%d:%.3f'%(c,t) for c,t in sorted(rows[lg])))
PY
