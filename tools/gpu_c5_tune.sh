# The wavefront kernel's run-time knobs on config 5 (refill threshold, parking limit, paths per workgroup, workgroups per CU, self-scheduling divisor, node-fetch form):
# tools/gpu_ab_env.py, one context, the two settings alternating, 10 rounds of 16 frames, GLRTX_NO_FEED=1.  -> profiles/r06_c5_tune.txt
L=opengl-raytracer_amd/lib/libglrtx.so
export GLRTX_NO_FEED=1
O="--contexts 1 --rounds 10 --frames 16 --config c5"
run() { echo "== $1 $2 -> $3"; timeout -k 10 200 python tools/gpu_ab_env.py $L $1 $2 $3 $O 2>&1 | tail -2; }
run GLRTX_REFILL_MIN 16 12
run GLRTX_REFILL_MIN 16 24
run GLRTX_REFILL_MIN 16 32
run GLRTX_SUSPEND_MAX 24 16
run GLRTX_SUSPEND_MAX 24 40
run GLRTX_SUSPEND_MAX 24 64
run GLRTX_BLOCK_PATHS 4096 2048
run GLRTX_WGS_PER_CU 4 3
run GLRTX_GSS_DIV 4096 8192
run GLRTX_PAIR_FETCH 1 2
