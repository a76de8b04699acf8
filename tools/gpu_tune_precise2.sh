L=opengl-raytracer_amd/lib/libglrtx.so
run() { echo "== $1: $2 vs $3"; timeout -k 10 120 python tools/gpu_ab_env.py $L $1 $2 $3 --contexts 3 --rounds 24 2>&1 | tail -3 | sed 's/images and rays identical//'; }
run GLRTX_GSS_DIV 4096 3072
run GLRTX_GSS_DIV 4096 5120
run GLRTX_SUSPEND_MAX 24 16
run GLRTX_SUSPEND_MAX 24 12
run GLRTX_SUSPEND_MAX 24 20
