#!/bin/bash
# kernel-trace only: per-kernel durations of the default bench command (fast)
set -e
TAG=${1:-kt}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/kt.log 2>&1
cat $OUT/kt/*/*_kernel_stats.csv
tail -1 $OUT/kt.log
