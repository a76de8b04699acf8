import sys, ctypes as C; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
device.lib_path = lambda: device.LIB_DIR / "libglrtx_stats.so"
sc, pr = scenes.config_headline()
d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"]); d.count_rays(True)
L = device.lib()
for depth in (1, 2, 8):
    p = dict(pr, max_depth=depth)
    d.reset_stats(); out = (C.c_ulonglong * 8)(); L.glrtx_debug_trav_stats(out)
    hh = (C.c_ulonglong * 16)(); L.glrtx_debug_trav_hist(hh)
    d.render_frames(p, [host.frame_seed(f) for f in range(4)]); d.sync()
    L.glrtx_debug_trav_stats(out); o = list(out); rays = d.stats().rays
    L.glrtx_debug_trav_hist(hh)
    print(f"depth {depth}: rays {rays} wave_iters {o[0]} active lanes/iter {o[1]/o[0]:.1f} fork_lane share {o[2]/o[1]:.3f} mixed_iters {o[4]/o[0]:.3f} iters/ray {o[1]/rays:.1f} distinct lines/iter {o[6]/o[0]:.1f} path-ray share {o[7]/o[1]:.3f}", flush=True)
