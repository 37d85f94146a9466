#!/bin/bash
# round 3, GPU session ag: where a wavefront's cycles go (parked / issue stall / issuing), headline passes and the division kernels
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ag; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU"
for cfg in "ntt22 --mode streams --streams 1" "ntt22 --mode streams --streams 1 --tile-logc 2" "open22" "batch16"; do
  set -- $cfg; wl=$1; shift
  tag=$(echo "$wl$*" | tr -d ' -')
  mkdir -p $OUT/$tag
  rocprofv3 --pmc $C -d $OUT/$tag/sq -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 40 --warmup 5 --samples 1 --no-cpu --no-verify $* > $OUT/$tag/sq.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $OUT/$tag $OUT/$tag/summary x > /dev/null 2>&1
  find $OUT/$tag -name "*.db" -delete 2>/dev/null; find $OUT/$tag -name "*.csv" -size +1M -delete 2>/dev/null
  echo "=== $tag"; grep -A9 "ntt_tile_kernel\|lindiv_" $OUT/$tag/summary.txt | grep -v "^--" | cut -c1-150 | head -44
done
