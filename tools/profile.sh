#!/bin/bash
# tools/profile.sh <workload> <tag> [extra bench args] -- rocprofv3 kernel trace + PMC passes of ONE bench.py command on the
# GPU box (run through gpurun); writes gpurun_out/prof_<tag>/ and the summary gpurun_out/prof_<tag>/summary.{txt,json}.
# The summary names itself "<tag>_rocprof.txt" (the `source` field bench.py quotes): commit it under profiles/ by THAT name.
# PMC passes are separate runs (TCC slot budget; never combined with trace domains -- gpurun refuses that anyway).
set -u
WL=$1; TAG=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --workload $WL --steps 100 --warmup 10 --samples 1 --no-cpu --no-verify $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1
python $ROOT/tools/rocprof_summary.py $OUT $OUT/summary "${TAG}_rocprof.txt" $WL > /dev/null
echo "command: $CMD" >> $OUT/summary.txt
# keep only the small artefacts (the sqlite traces are large)
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -name "*.csv" -size +1M -delete 2>/dev/null
tail -40 $OUT/summary.txt
