#!/bin/bash
# Round 4 closing measurements: profiles of the headline and of configs 3 and 5 (part 1), the two bench lines, the configs table and the 600-seed fuzz soak (part 2).
# Usage (GPU box): tools/gpu_r04_final.sh 1|2    -- two calls, each inside one gpurun time limit
OUT=gpurun_out/r04_final; mkdir -p $OUT
if [ "${1:-1}" = 1 ]; then
  bash tools/profile.sh r04 headline > $OUT/prof_headline.log 2>&1; echo "profile headline done"
  bash tools/profile.sh r04_c3 c3 > $OUT/prof_c3.log 2>&1; echo "profile c3 done"
  bash tools/profile.sh r04_c5 c5 "--bvh lbvh" > $OUT/prof_c5.log 2>&1; echo "profile c5 done"
else
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_n1_steps20.json 2> $OUT/bench_n1_steps20.err; echo "bench 20 done"
  timeout -k 10 300 python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench default done"
  timeout -k 10 300 python3 bench.py --config c3 --no-single > $OUT/bench_c3.json 2> $OUT/bench_c3.err; echo "bench c3 done"
  timeout -k 10 300 python3 bench.py --config c5 --bvh lbvh --no-single > $OUT/bench_c5_lbvh.json 2> $OUT/bench_c5_lbvh.err; echo "bench c5 done"
  timeout -k 10 300 python3 tools/gpu_configs.py > $OUT/configs.txt 2>&1; echo "configs done"
  GLRT_FUZZ_SEEDS=600 timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > $OUT/fuzz600.txt 2>&1; tail -2 $OUT/fuzz600.txt
fi
git rev-parse HEAD 2>/dev/null > $OUT/head.txt || true
