#!/bin/bash
# Round 4: the fourth load of a node record under the fork arm's exec mask (new) against all lanes loading it (base = -DGLRTX_FOURTH_LOAD_ALL_LANES), over tree sizes.
# Usage (GPU box): tools/gpu_ab_load4.sh > gpurun_out/ab_load4_sizes.txt
L=opengl-raytracer_amd/lib
for n in 4000 12000 24000 48000; do
  echo "#### rand:$n"
  timeout -k 10 200 python3 tools/gpu_abx.py --config rand:$n --frames 16 --rounds 4 --repeat 2 \
    base2=$L/libglrtx_base.so,GLRTX_PAIR_FETCH=2 new2=$L/libglrtx.so,GLRTX_PAIR_FETCH=2 base1=$L/libglrtx_base.so,GLRTX_PAIR_FETCH=1 base0=$L/libglrtx_base.so,GLRTX_PAIR_FETCH=0 new0=$L/libglrtx.so,GLRTX_PAIR_FETCH=0 2>&1 | grep "^=="
done
