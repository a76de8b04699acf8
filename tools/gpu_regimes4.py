"""Diagnostic: which phase of the trip differs between a fast and a slow context of one binary (tools/gpu_ab_inproc.py: up to 3 % apart)?  N contexts of the
-DGLRTX_PHASE_STATS build in one process; per context kernel ms per frame and the shader clocks summed per phase.   python tools/gpu_regimes4.py [N]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
from glrt_amd import device, host, scenes  # noqa: E402

device.lib_path = lambda: device.LIB_DIR / os.environ.get("GLRTX_PHASE_LIB", "libglrtx_phase.so")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = 20
sc, pr = scenes.CONFIGS["headline"]()
L = device.lib(); out = (C.c_ulonglong * 8)()
ds = []
for i in range(N):
    d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"]); d.count_rays(False)
    d.render_frames(pr, [host.frame_seed(i_) for i_ in range(frames)]); d.sync(); L.glrtx_debug_phase_cycles(out)
    ds.append(d)
names = ["top-up", "traverse", "wait(trav)", "shade", "wait(shade)"]
print("ctx   ms/frame   " + "  ".join(f"{n:>11s}" for n in names) + "   (Mcycles summed over workgroups, mean of 4 launches)")
r = 1
for i, d in enumerate(ds):
    ms, ph = [], []
    for _ in range(4):
        d.render_frames(pr, [host.frame_seed(frames * r + i_) for i_ in range(frames)]); d.sync(); r += 1
        L.glrtx_debug_phase_cycles(out)
        ms.append(d.stats().kernel_ms_last / frames); ph.append([float(x) for x in list(out)[:5]])
    ph = np.asarray(ph).mean(0) / 1e6
    print(f"{i:3d}   {np.median(ms):.4f}    " + "  ".join(f"{v:11.1f}" for v in ph), flush=True)
