#!/bin/bash
OUT=gpurun_out/r04_pair2; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.txt 2>&1; tail -4 $OUT/gpu_tests.txt
L=opengl-raytracer_amd/lib/libglrtx.so
for cfg in headline c4 c5 c2; do
  fr=24; [ $cfg = c5 ] && fr=16; [ $cfg = c4 ] && fr=8
  timeout -k 10 300 python3 tools/gpu_abx.py --config $cfg --frames $fr --rounds 4 --repeat 2 lane=$L,GLRTX_PAIR_FETCH=0 pair=$L,GLRTX_PAIR_FETCH=1 > $OUT/ab_$cfg.txt 2>&1; grep "==" $OUT/ab_$cfg.txt
done
