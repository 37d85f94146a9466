O=gpurun_out/r02z; mkdir -p $O
export TMPDIR=/tmp
for c in 14 15; do
  RONK_MSM_C=$c timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c$c -o msm -- python bench.py --workload msm20 --no-cpu --no-verify --steps 2 --samples 1 --warmup 1 > $O/bench_c$c.json 2>$O/err_c$c
  f=$(find $O/prof_c$c -name "*kernel_stats.csv" | head -1); echo "== c=$c $f"; head -14 "$f" | cut -d, -f1-8 | cut -c1-200
done
find $O -name "*.csv" | head; du -sh $O
