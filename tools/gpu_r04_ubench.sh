#!/bin/bash
# Round 4, run on the GPU box: the two issue-rate micro-benchmarks with sustained trains and in-kernel clocks, then the
# driver-style bench line, the per-step clocks and the traversal statistics of the kernel as it stands.
set -e
OUT=gpurun_out/r04_ubench
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 420 tools/ubench/ta $OUT/ta.jsonl 2.0 > $OUT/ta.txt 2>&1
echo "ta done"
timeout -k 10 420 tools/ubench/valu $OUT/valu.jsonl > $OUT/valu.txt 2>&1
echo "valu done"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
echo "bench done"
timeout -k 10 200 python3 tools/gpu_steptime.py headline 16 > $OUT/steptime.txt 2>&1
timeout -k 10 200 python3 tools/gpu_travstats.py headline 24 > $OUT/travstats.txt 2>&1
echo "diag done"
tail -3 $OUT/ta.txt; tail -2 $OUT/valu.txt; cat $OUT/bench_steps20.json | head -c 600; cat $OUT/steptime.txt $OUT/travstats.txt
