"""Where a workgroup's time goes (libglrtx_phase.so, -DGLRTX_PHASE_STATS): shader clocks of the top-up / traverse / shade phases and of the barrier waits behind them, summed
over the workgroups of one launch; the refill sections and stepping blocks of wave 0 inside the traverse phase.
    [GLRTX_PHASE_CONFIG=headline|c2..c5] [GLRTX_PHASE_LIB=libglrtx_phase.so] python tools/gpu_phase.py [frames per launch = 16]
"""
import sys, ctypes as C; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
import os
device.lib_path = lambda: device.LIB_DIR / os.environ.get("GLRTX_PHASE_LIB", "libglrtx_phase.so")
sc, pr = scenes.CONFIGS[os.environ.get("GLRTX_PHASE_CONFIG", "headline")]()
d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"])
L = device.lib(); out = (C.c_ulonglong * 8)()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
d.render_frames(pr, [host.frame_seed(f) for f in range(B)]); d.sync(); L.glrtx_debug_phase_cycles(out)
d.render_frames(pr, [host.frame_seed(B + f) for f in range(B)]); d.sync(); L.glrtx_debug_phase_cycles(out)
o = np.array(list(out)[:5], float); print("frames per launch", B, "ms per frame", d.stats().kernel_ms_last / B)
names = ["top-up/generate", "traverse", "wait after traverse", "shade", "wait after shade"]
for n, v in zip(names, o): print(f"{n:22s} {v/o.sum()*100:5.1f} %   ({v/1e6:.1f} Mcycles summed over workgroups)")
o2 = list(out)
print(f"refill sections (wave 0 of every workgroup): {o2[5]/max(o[1],1)*100:.1f} % of the traverse phase, {o2[6]} events, {o2[5]/max(o2[6],1):.0f} cycles each")
print(f"stepping blocks (wave 0 of every workgroup): {o2[7]/max(o[1],1)*100:.1f} % of the traverse phase")
