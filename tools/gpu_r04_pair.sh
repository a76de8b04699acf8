#!/bin/bash
# Round 4: the pair-cooperative node fetch against one record per lane -- parity first, then the A/B timing
OUT=gpurun_out/r04_pair; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or oracle or variants or parked" > $OUT/parity.txt 2>&1; tail -5 $OUT/parity.txt
timeout -k 10 400 python3 tools/gpu_abx.py --frames 24 --rounds 5 --repeat 2 pair=opengl-raytracer_amd/lib/libglrtx.so lane=opengl-raytracer_amd/lib/libglrtx_nopair.so > $OUT/ab_headline.txt 2>&1; cat $OUT/ab_headline.txt
timeout -k 10 400 python3 tools/gpu_abx.py --config c5 --frames 16 --rounds 4 --repeat 2 pair=opengl-raytracer_amd/lib/libglrtx.so lane=opengl-raytracer_amd/lib/libglrtx_nopair.so > $OUT/ab_c5.txt 2>&1; cat $OUT/ab_c5.txt
