L=opengl-raytracer_amd/lib/libglrtx.so
run() { echo "== $1: $2 vs $3"; timeout -k 10 120 python tools/gpu_ab_env.py $L $1 $2 $3 --contexts 2 --rounds 20 2>&1 | tail -2 | sed 's/images and rays identical//'; }
run GLRTX_REFILL_MIN 16 12
run GLRTX_REFILL_MIN 16 14
run GLRTX_REFILL_MIN 16 20
run GLRTX_SUSPEND_MAX 24 16
run GLRTX_SUSPEND_MAX 24 32
run GLRTX_SUSPEND_MAX 24 40
run GLRTX_GSS_DIV 4096 2048
run GLRTX_GSS_DIV 4096 8192
run GLRTX_GSS_DIV 4096 16384
run GLRTX_PAIR_FETCH 2 0
run GLRTX_PAIR_FETCH 2 1
