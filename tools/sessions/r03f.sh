#!/bin/bash
# round 3, GPU session f: rocprofv3 evidence (kernel trace + PMC) for the headline and the multiply
OUT=gpurun_out/r03f; mkdir -p $OUT
timeout 500 bash tools/profile.sh ntt22 1stream --mode streams --streams 1 > $OUT/prof_1stream.txt 2>&1
timeout 500 bash tools/profile.sh ntt22 many > $OUT/prof_many.txt 2>&1
timeout 500 bash tools/profile.sh mul22 mul22 > $OUT/prof_mul22.txt 2>&1
timeout 500 bash tools/profile.sh batch16 batch16 > $OUT/prof_batch16.txt 2>&1
for t in 1stream many mul22 batch16; do cp gpurun_out/prof_$t/summary.txt $OUT/summary_$t.txt; cp gpurun_out/prof_$t/summary.json $OUT/summary_$t.json; done
head -12 $OUT/summary_1stream.txt; head -12 $OUT/summary_many.txt; head -12 $OUT/summary_mul22.txt; head -10 $OUT/summary_batch16.txt
