"""Diagnostic: phase cycles of the headline config against the same scene with its four conductor spheres made diffuse --
an upper bound on what the conductor branch (and the divergence it causes in the shade phase) costs."""
import sys, ctypes as C, os; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
device.lib_path = lambda: device.LIB_DIR / os.environ.get("GLRTX_PHASE_LIB", "libglrtx_phase.so")
L = None
for variant in ("headline", "all diffuse"):
    orig = scenes.conductor
    if variant == "all diffuse": scenes.conductor = lambda eta, kappa, alpha: scenes.diffuse((0.7, 0.5, 0.3))
    sc, pr = scenes.config_headline()
    scenes.conductor = orig
    d = device.Device(); d.upload_scene(sc); d.resize(1920, 1080)
    L = device.lib(); out = (C.c_ulonglong * 8)()
    B = 24
    d.count_rays(True); d.reset_stats(); d.render_frames(pr, [host.frame_seed(f) for f in range(B)]); d.sync(); st = d.stats(); rays = int(st.rays); d.count_rays(False)
    L.glrtx_debug_phase_cycles(out)
    d.render_frames(pr, [host.frame_seed(B + f) for f in range(B)]); d.sync(); L.glrtx_debug_phase_cycles(out)
    o = np.array(list(out)[:5], float)
    print(f"{variant:12s}: {d.stats().kernel_ms_last / B:.4f} ms per frame, {rays / B / 1e6:.2f} M reference rays per frame; Mcycles summed over workgroups: "
          + "  ".join(f"{n} {v/1e6:.0f}" for n, v in zip(["top-up", "traverse", "wait", "shade", "wait"], o)), flush=True)
    d.close()
