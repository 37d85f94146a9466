O=gpurun_out/r02ac; mkdir -p $O
for ch in 8 12 16 24 32 64 128; do RONK_MSM_CH=$ch timeout 300 python bench.py --workload msm20 --log2n 20 --no-cpu --no-verify --steps 5 --samples 3 > $O/msm20_ch$ch.json 2>> $O/err; done
for ch in 8 16 32; do for c in 14 16; do RONK_MSM_C=$c RONK_MSM_CH=$ch timeout 300 python bench.py --workload msm20 --log2n 20 --no-cpu --no-verify --steps 5 --samples 3 > $O/msm20_c${c}_ch$ch.json 2>> $O/err; done; done
for ch in 4 8 16 32; do RONK_MSM_CH=$ch timeout 300 python bench.py --workload msm20 --log2n 16 --no-cpu --no-verify --steps 5 --samples 3 > $O/msm16_ch$ch.json 2>> $O/err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02ac/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'Mpoints/s %.2f'%(d['value']/1e6), 'ms/step %.3f'%d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
