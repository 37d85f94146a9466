O=gpurun_out/r02aa; mkdir -p $O; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o msm -- python bench.py --workload msm20 --no-cpu --no-verify --steps 5 --samples 1 --warmup 1 > $O/bench.json 2>$O/err
python tools/rocpd_stats.py "$O/prof/**/*.db" | cut -c1-130 | tee $O/rocprof_msm20.txt
cat $O/bench.json | cut -c1-200
find $O -name "*.db" -delete
