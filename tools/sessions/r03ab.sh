#!/bin/bash
# round 3, GPU session ab: lane-scan division forms -- full parity run + kernel split and VALU counters per form
OUT=gpurun_out/r03ab; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "scan_onepass_variants or horner_scan or kzg" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
for m in 0 8l 16l 8d; do
  RONK_LINDIV=$m bash tools/profile.sh open22 lindiv_$m > /dev/null 2>&1
  cp gpurun_out/prof_lindiv_$m/summary.txt $OUT/summary_$m.txt
  echo "=== $m"; grep -E "lindiv|chunk_sum|VALU|FETCH|WRITE|valu|fetch|write" $OUT/summary_$m.txt | head -14 | cut -c1-200
done
