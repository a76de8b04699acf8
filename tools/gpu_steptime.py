"""Diagnostic (GPU box, lib built with -DGLRTX_STEP_TIMING: make -C opengl-raytracer_amd diag): what a lane's traversal step costs in shader
clocks inside trav_steps_asm, and how much of that is the s_waitcnt behind the node fetch.  Usage: gpu_steptime.py [config] [frames per launch]"""
import sys, ctypes as C; sys.path.insert(0, '.'); sys.path.insert(0, 'opengl-raytracer_amd/python')
from glrt_amd import scenes, device, host
device.lib_path = lambda: device.LIB_DIR / "libglrtx_steptime.so"
cfg = sys.argv[1] if len(sys.argv) > 1 else "headline"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sc, pr = scenes.CONFIGS[cfg]()
d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"])
L = device.lib(); out = (C.c_ulonglong * 4)()
d.render_frames(pr, [host.frame_seed(f) for f in range(B)]); d.sync(); L.glrtx_debug_step_timing(out)
d.render_frames(pr, [host.frame_seed(B + f) for f in range(B)]); d.sync(); L.glrtx_debug_step_timing(out)
n, t, w = out[0], out[1], out[2]
print(f"{cfg}: {B} frames per launch, {d.stats().kernel_ms_last / B:.3f} ms per frame (instrumented)")
print(f"lane-steps {n}: {t / max(n, 1):.0f} clk per step as the wave sees it, {w / max(n, 1):.0f} clk of it in s_waitcnt vmcnt(0) ({100.0 * w / max(t, 1):.1f} %)")
