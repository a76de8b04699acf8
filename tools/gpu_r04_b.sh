#!/bin/bash
set -e
OUT=gpurun_out/r04_b; rm -rf $OUT; mkdir -p $OUT
for c in 4:0:64 4:7:23 4:0:1 4:0:23; do timeout -k 5 60 tools/ubench/ta /dev/null 0.5 $c >> $OUT/ta_residency.txt 2>&1; done
cat $OUT/ta_residency.txt
timeout -k 10 200 python3 tools/gpu_travstats.py headline 8 > $OUT/travstats.txt 2>&1; cat $OUT/travstats.txt
timeout -k 10 400 python3 tools/gpu_replay.py headline 4 5 > $OUT/replay.txt 2>&1; cat $OUT/replay.txt
bash tools/profile_ta.sh r04 headline rest > $OUT/profile_ta.txt 2>&1; tail -70 $OUT/profile_ta.txt
